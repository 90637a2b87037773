/* A plain-C client of include/sga_hip.h: no Python, no torch in this process.
 *
 *   sga_client <weights.bin> <x.bin> <out.bin> C B H W its lambda seed
 *
 * weights.bin: the members of sga_weights in declaration order, interleaved per layer as read below
 * (ga: kernel, bias, [beta, gamma] x4/x3; gs: same; ha: kernel, [bias]; hs: kernel, bias; eb: matrix,
 * bias, [factor]), float32, each preceded by its element count (int64).  x.bin: [B,H,W,3] float32.  out.bin:
 * metrics [B,7], y_hat, z_hat as written by sga_run.  tests/test_gpu_c_abi.py builds this with gcc,
 * runs it and compares the output bit-for-bit with the Python host calling the same library. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "sga_hip.h"

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d hip error %d\n", __FILE__, __LINE__, (int)_e); return 2; } } while (0)
#define CHECK_SGA(e) do { int _s = (e); if (_s != 0) { char m[256] = {0}; sga_last_error(h, m, sizeof m); fprintf(stderr, "%s:%d sga status %d %s\n", __FILE__, __LINE__, _s, m); return 3; } } while (0)

static float* read_tensor(FILE* f, int64_t* n_out) {
  int64_t n = 0;
  if (fread(&n, sizeof n, 1, f) != 1) return NULL;
  float* p = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  if (n > 0 && fread(p, sizeof(float), (size_t)n, f) != (size_t)n) { free(p); return NULL; }
  if (n_out) *n_out = n;
  return p;
}

int main(int argc, char** argv) {
  if (argc != 11) { fprintf(stderr, "usage: %s weights.bin x.bin out.bin C B H W its lambda seed\n", argv[0]); return 1; }
  const int C = atoi(argv[4]), B = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), its = atoi(argv[8]);
  const float lambda = (float)atof(argv[9]);
  const uint64_t seed = strtoull(argv[10], NULL, 10);
  if (sga_abi_version() != SGA_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

  /* ---- weights --------------------------------------------------------------------------------- */
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  sga_weights w;
  memset(&w, 0, sizeof w);
  int64_t n;
#define NEXT(dst) do { (dst) = read_tensor(f, &n); if (!(dst)) { fprintf(stderr, "weights.bin truncated at %s\n", #dst); return 1; } } while (0)
  for (int k = 0; k < 4; ++k) { NEXT(w.ga_kernel[k]); NEXT(w.ga_bias[k]); if (k < 3) { NEXT(w.ga_beta[k]); NEXT(w.ga_gamma[k]); } }
  for (int k = 0; k < 4; ++k) { NEXT(w.gs_kernel[k]); NEXT(w.gs_bias[k]); if (k < 3) { NEXT(w.gs_beta[k]); NEXT(w.gs_gamma[k]); } }
  for (int k = 0; k < 3; ++k) { NEXT(w.ha_kernel[k]); if (k < 2) NEXT(w.ha_bias[k]); }
  for (int k = 0; k < 3; ++k) { NEXT(w.hs_kernel[k]); NEXT(w.hs_bias[k]); }
  for (int k = 0; k < 4; ++k) { NEXT(w.eb_matrix[k]); NEXT(w.eb_bias[k]); if (k < 3) NEXT(w.eb_factor[k]); }
  fclose(f);

  sga_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.num_filters = C; cfg.max_batch = B; cfg.max_height = H; cfg.max_width = W;
  cfg.precision = SGA_PRECISION_F32_MFMA;
  sga_handle* h = NULL;
  int st = sga_create(&h, &cfg, &w);
  if (st != 0) { fprintf(stderr, "sga_create -> %d\n", st); return 3; }

  int yh, yw, zh, zw;
  CHECK_SGA(sga_latent_shape(h, H, W, &yh, &yw, &zh, &zw));
  const size_t nx = (size_t)B * H * W * 3, ny = (size_t)B * yh * yw * C, nz = (size_t)B * zh * zw * C, nm = (size_t)B * 7;

  float* hx = (float*)malloc(nx * sizeof(float));
  f = fopen(argv[2], "rb");
  if (!f || fread(hx, sizeof(float), nx, f) != nx) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
  fclose(f);

  /* device memory from the plain HIP allocator: the ABI takes raw device pointers */
  float *dx, *dy, *dz, *dm;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_HIP(hipMalloc((void**)&dx, nx * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dy, ny * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dz, nz * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&dm, nm * sizeof(float)));
  CHECK_HIP(hipMemcpy(dx, hx, nx * sizeof(float), hipMemcpyHostToDevice));

  CHECK_SGA(sga_run(h, dx, B, H, W, lambda, 1.0f / (float)B, its, 0.005, 1e-3, 700, 0.5, seed,
                    NULL, NULL, dy, dz, dm, NULL, (void*)stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  float* out = (float*)malloc((nm + ny + nz) * sizeof(float));
  CHECK_HIP(hipMemcpy(out, dm, nm * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out + nm, dy, ny * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out + nm + ny, dz, nz * sizeof(float), hipMemcpyDeviceToHost));
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out, sizeof(float), nm + ny + nz, f) != nm + ny + nz) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
  fclose(f);
  printf("ok: B=%d y=[%d,%d,%d] z=[%d,%d,%d] est_bpp[0]=%.6f psnr[0]=%.4f\n", B, yh, yw, C, zh, zw, out[4], out[1]);

  /* error behaviour of the ABI */
  if (sga_run(h, dx, B + 1, H, W, lambda, 1.f, its, 0.005, 1e-3, 700, 0.5, seed, NULL, NULL, dy, dz, dm, NULL, stream) != SGA_ERR_BAD_SHAPE) return 4;
  if (sga_run(NULL, dx, B, H, W, lambda, 1.f, its, 0.005, 1e-3, 700, 0.5, seed, NULL, NULL, dy, dz, dm, NULL, stream) != SGA_ERR_BAD_ARG) return 4;
  CHECK_SGA(sga_destroy(h));
  return 0;
}
