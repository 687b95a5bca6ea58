/* Exhaustive proof that sample_sky's division-free x/255 (kernels.hip div255: q = x*rc; r = fma(-q,255,x);
 * q' = fma(r,rc,q) with rc = RN(1/255)) returns the IEEE-754 quotient for EVERY binary32 x in [0, 256] — the cube-map
 * filter only produces values in [0, 255].  Usage: gcc -O2 -ffp-contract=off check_div255.c -lm -lpthread && ./a.out [stride]
 * (stride 1 = all 1.13e9 values, ~5 s on 8 threads; tests/test_oracle.py runs a stride of 61). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
static float rc;
static uint32_t stride;
static uint64_t bad[8];
static uint32_t first_bad[8];
static void* run(void* arg) {
  int t = (int)(intptr_t)arg;
  const uint32_t hi = 0x43800000u;  /* 256.0f */
  uint64_t nb = 0;
  for (uint64_t b64 = (uint64_t)t * stride; b64 <= hi; b64 += 8ull * stride) {
    const uint32_t b = (uint32_t)b64;
    float x; memcpy(&x, &b, 4);
    float q1 = x * rc;
    float r = fmaf(-q1, 255.0f, x);
    float q2 = fmaf(r, rc, q1);
    float ref = x / 255.0f;
    if (memcmp(&q2, &ref, 4) != 0) { if (!nb) first_bad[t] = b; nb++; }
  }
  bad[t] = nb;
  return 0;
}
int main(int argc, char** argv) {
  stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
  if (stride == 0) stride = 1;
  rc = (float)(1.0 / 255.0);
  pthread_t th[8];
  for (int t = 0; t < 8; t++) pthread_create(&th[t], 0, run, (void*)(intptr_t)t);
  uint64_t tot = 0;
  for (int t = 0; t < 8; t++) { pthread_join(th[t], 0); tot += bad[t]; if (bad[t]) printf("thread %d first bad bits %08x\n", t, first_bad[t]); }
  printf("rc bits %08x mismatches %llu\n", *(uint32_t*)&rc, (unsigned long long)tot);
  return tot != 0;
}
