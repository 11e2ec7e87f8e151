/* quad_oracle_f32.c -- TEST INFRASTRUCTURE: the SAME restatement (quad_oracle.c, included below) compiled with every
 * `double` of its own text read as `float`: float storage and float arithmetic (gcc -fsingle-precision-constant makes
 * the literals float too; libm calls that gcc does not narrow go through double and are rounded back).  Only the batch
 * drivers (qo_batch_rollout, qo_batch_ppo: plain array arguments, float arrays here) are called on this build -- it is
 * the fp32 CPU baseline SURVEY.md section 8(d)(ii) asks for next to the fp64 one; parity is judged on the fp64 build. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define double float
#include "quad_oracle.c"
