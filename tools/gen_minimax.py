"""Coefficients of the polynomial kernels in csrc/qenv_fastmath.hpp (lean fp64 atan2 / asin / sin / cos).
Near-minimax fits by Chebyshev interpolation in 60-digit arithmetic (mpmath.chebyfit), printed as C initialisers with
the fit's maximum error.  Run:  python tools/gen_minimax.py"""
import mpmath as mp

mp.mp.dps = 60


def fit(name, f, a, b, n, what):
    c, err = mp.chebyfit(f, [a, b], n, error=True)  # c[0] = highest power
    c = c[::-1]
    print(f"// {what}: degree {n - 1} on [{mp.nstr(a, 6)}, {mp.nstr(b, 6)}], max fit error {mp.nstr(err, 3)}")
    print(f"static constexpr double {name}[{n}] = {{" + ", ".join(mp.nstr(x, 20) for x in c) + "};")


T8 = mp.tan(mp.pi / 8)
# atan(t) = t * P(z), z = t^2, |t| <= tan(pi/8)
fit("ATAN_P", lambda z: mp.atan(mp.sqrt(z)) / mp.sqrt(z) if z > 0 else mp.mpf(1), mp.mpf(0), T8 * T8 * mp.mpf("1.0001"), 12, "atan(t)/t in z = t^2")
# asin(s) = s + s z R(z), z = s^2 <= 0.25
fit("ASIN_R", lambda z: (mp.asin(mp.sqrt(z)) - mp.sqrt(z)) / (mp.sqrt(z) * z) if z > 0 else mp.mpf(1) / 6, mp.mpf(0), mp.mpf("0.2501"), 12, "(asin(s) - s)/s^3 in z = s^2")
Q = (mp.pi / 4) ** 2 * mp.mpf("1.0001")
# sin(r) = r + r z S(z), cos(r) = 1 - z/2 + z^2 C(z), z = r^2 <= (pi/4)^2
fit("SIN_S", lambda z: (mp.sin(mp.sqrt(z)) - mp.sqrt(z)) / (mp.sqrt(z) * z) if z > 0 else -mp.mpf(1) / 6, mp.mpf(0), Q, 6, "(sin(r) - r)/r^3 in z = r^2")
fit("COS_C", lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / (z * z) if z > 0 else mp.mpf(1) / 24, mp.mpf(0), Q, 6, "(cos(r) - 1 + z/2)/z^2 in z = r^2")
print("// pi/2 split for Cody-Waite reduction: hi has 33 significant bits (k * hi exact for |k| < 2^20)")
import struct
pio2 = mp.pi / 2
hi = float(pio2)
hi = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", hi))[0] & ~((1 << 20) - 1)))[0]
mid = float(pio2 - mp.mpf(hi))
mid_t = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", mid))[0] & ~((1 << 20) - 1)))[0]
lo = float(pio2 - mp.mpf(hi) - mp.mpf(mid_t))
print(f"static constexpr double PIO2_HI = {hi!r}, PIO2_MID = {mid_t!r}, PIO2_LO = {lo!r};")
