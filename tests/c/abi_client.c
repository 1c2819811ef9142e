/* A plain-C client of include/summa_gpu.h, the way a cgo / Rust-FFI / JNI binding sees the library:
 * compiled with gcc -std=c99 and linked against libsumma_gpu.so by tests/test_host_logic.py.  Without a GPU
 * it exercises only host-side entry points and the loud-failure contract. */
#include <stdio.h>
#include <string.h>

#include "summa_gpu.h"

int main(void) {
  unsigned char one[32] = {0xfb, 0xff, 0xff, 0x4f, 0x1c, 0x34, 0x96, 0xac, 0x29, 0xcd, 0x60, 0x9f, 0x95, 0x76, 0xfc, 0x36,
                           0x2e, 0x46, 0x79, 0x78, 0x6f, 0xa3, 0x6e, 0x66, 0x2f, 0xdf, 0x07, 0x9a, 0xc1, 0x77, 0x0a, 0x0e};
  unsigned char g2[128], sum[64], pts[128];
  const char* v = sg_version();
  if (!v || strncmp(v, "summa_gpu", 9) != 0) return 1;
  if (sg_g2_generator_mul(one, g2) != SG_OK) return 2;           /* host code: works without a device */
  if (g2[0] != 0x26 || g2[1] != 0x20) return 3;                  /* low limb 0x8e83b5d102bc2026 of x.c0 */
  memset(pts, 0, sizeof pts);
  if (sg_g1_sum_affine(pts, 2, sum) != SG_OK) return 4;          /* identity + identity */
  for (int i = 0; i < 64; i++)
    if (sum[i]) return 5;
  if (sg_device_count() <= 0) {                                  /* no GPU: compute calls must fail loudly */
    unsigned char out[64];
    if (sg_msm_g1(one, pts, 1, out) == SG_OK) return 6;
    if (!sg_last_error() || !sg_last_error()[0]) return 7;
  }
  printf("abi client ok: %s\n", v);
  return 0;
}
