"""Parse the rocprofv3 --pmc CSVs written by tools/pmc_traffic.sh: per-kernel average counter
values, the calibration factor of each counter on the known-size copy, and the calibrated HBM
bytes per launch of the env kernels."""
import csv, glob, json, os, sys, collections

def load(d):
    """{kernel_name: [counter values per dispatch]} from a pass directory."""
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out

def avg(v):
    return sum(v) / max(1, len(v))

root = sys.argv[1]
res = {}
calib_bytes = 13 * 16 * 1024 * 1024 * 4
factor = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    k = load(os.path.join(root, "calib_" + c))
    vals = [v for name, vs in k.items() if "k_copy_rows" in name for v in vs]
    if vals:
        # counters are in KiB (rocprofv3 derived metric); factor = true bytes / reported bytes
        factor[c] = calib_bytes / (avg(vals) * 1024.0)
        print(f"calibration {c}: reported {avg(vals) * 1024 / 1e6:.1f} MB per launch vs true {calib_bytes / 1e6:.1f} MB -> factor {factor[c]:.3f}")
for tag in ("step", "fused", "big"):
    entry = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        k = load(os.path.join(root, f"{tag}_{c}"))
        for name, vs in k.items():
            if "k_step" in name or "k_rollout" in name:
                steady = vs[len(vs) // 2:]
                entry.setdefault(name, {})[c] = avg(steady) * 1024.0
                entry[name]["launches"] = len(vs)
    for name, e in entry.items():
        raw = e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)
        cal = e.get("FETCH_SIZE", 0) * factor.get("FETCH_SIZE", 1) + e.get("WRITE_SIZE", 0) * factor.get("WRITE_SIZE", 1)
        print(f"{tag}: {name[:60]} launches={e['launches']} fetch={e.get('FETCH_SIZE', 0) / 1e6:.2f} MB write={e.get('WRITE_SIZE', 0) / 1e6:.2f} MB raw={raw / 1e6:.2f} MB calibrated={cal / 1e6:.2f} MB per launch")
        res[f"{tag}:{name}"] = dict(fetch_bytes=e.get("FETCH_SIZE", 0), write_bytes=e.get("WRITE_SIZE", 0), calibrated_bytes=cal)
res["factors"] = factor
json.dump(res, open(os.path.join(root, "traffic_raw.json"), "w"), indent=1)
