// Plain C++ host over the C ABI of include/dsir.h — no Python, no torch: what a maintainer of a C/C++ pipeline would
// write to register point-cloud pairs with libdsir.so.
//
//   cabi_register <weights.bin> <pairs.bin> <out.bin> [n_iter]
//
// weights.bin : int32 nkeys, then per key { int32 name_len, name bytes, int32 ndim, int64 dims[ndim], float32 data }
//               (ndim == -1: a key without float payload, e.g. '*.num_batches_tracked': loaded with a NULL pointer)
// pairs.bin   : int32 P, int32 N, int32 feat_len, float32 src[P][N][feat_len], float32 ref[P][N][feat_len]
// out.bin     : float32 transforms[P][n_iter][3][4], int32 invalid[P]
// tools/export_cabi_inputs.py writes the two inputs; tests/test_cabi_example.py compares out.bin with the Python host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dsir.h"

#define HIP_CHECK(x)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define DSIR_CHECK(ctx, x)                                                          \
  do {                                                                              \
    if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, dsir_last_error(ctx)); return 3; } \
  } while (0)

static bool read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s weights.bin pairs.bin out.bin [n_iter]\n", argv[0]); return 1; }
  const int n_iter = argc > 4 ? atoi(argv[4]) : 5;

  FILE* fp = fopen(argv[2], "rb");
  if (!fp) { perror(argv[2]); return 1; }
  int32_t P = 0, N = 0, C = 0;
  if (!read_exact(fp, &P, 4) || !read_exact(fp, &N, 4) || !read_exact(fp, &C, 4)) return 1;
  std::vector<float> src((size_t)P * N * C), ref((size_t)P * N * C);
  if (!read_exact(fp, src.data(), src.size() * 4) || !read_exact(fp, ref.data(), ref.size() * 4)) return 1;
  fclose(fp);

  dsir_cfg cfg{};
  cfg.feat_len = C; cfg.num_knn = 16; cfg.num_layers = 4;
  const int d_out[4] = {16, 64, 128, 256};
  for (int i = 0; i < 4; ++i) { cfg.sub_sampling_ratio[i] = 4; cfg.d_out[i] = d_out[i]; }
  cfg.out_feat_dim = 64; cfg.num_classes = 19; cfg.max_points = N < 1024 ? 1024 : N; cfg.max_pairs = P;
  cfg.pipeline = DSIR_PIPELINE_ALIGN;
  dsir_ctx* ctx = nullptr;
  if (dsir_create(0, &cfg, &ctx) != 0) { fprintf(stderr, "dsir_create: %s\n", dsir_last_error(nullptr)); return 3; }

  FILE* fw = fopen(argv[1], "rb");
  if (!fw) { perror(argv[1]); return 1; }
  int32_t nkeys = 0;
  if (!read_exact(fw, &nkeys, 4)) return 1;
  for (int k = 0; k < nkeys; ++k) {
    int32_t len = 0, ndim = 0;
    if (!read_exact(fw, &len, 4)) return 1;
    std::string name(len, '\0');
    if (!read_exact(fw, &name[0], len) || !read_exact(fw, &ndim, 4)) return 1;
    if (ndim < 0) { DSIR_CHECK(ctx, dsir_load_weight(ctx, name.c_str(), nullptr, nullptr, 0)); continue; }
    std::vector<int64_t> dims(ndim);
    size_t numel = 1;
    if (ndim && !read_exact(fw, dims.data(), (size_t)ndim * 8)) return 1;
    for (int64_t d : dims) numel *= (size_t)d;
    std::vector<float> data(numel);
    if (!read_exact(fw, data.data(), numel * 4)) return 1;
    DSIR_CHECK(ctx, dsir_load_weight(ctx, name.c_str(), data.data(), dims.data(), ndim));
  }
  fclose(fw);
  DSIR_CHECK(ctx, dsir_finalize_weights(ctx));

  float *d_src = nullptr, *d_ref = nullptr, *d_T = nullptr;
  int32_t* d_inv = nullptr;
  HIP_CHECK(hipMalloc((void**)&d_src, src.size() * 4));
  HIP_CHECK(hipMalloc((void**)&d_ref, ref.size() * 4));
  HIP_CHECK(hipMalloc((void**)&d_T, (size_t)P * n_iter * 12 * 4));
  HIP_CHECK(hipMalloc((void**)&d_inv, (size_t)P * 4));
  HIP_CHECK(hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_ref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice));

  dsir_pair_batch in{};
  in.pairs = P; in.n_src = N; in.n_ref = N; in.points_src = d_src; in.points_ref = d_ref;   // pyramids NULL: built on device
  dsir_pair_result out{};
  out.transforms = d_T; out.invalid = d_inv;
  DSIR_CHECK(ctx, dsir_register(ctx, &in, n_iter, &out));
  DSIR_CHECK(ctx, dsir_sync(ctx));

  std::vector<float> T((size_t)P * n_iter * 12);
  std::vector<int32_t> inv(P);
  HIP_CHECK(hipMemcpy(T.data(), d_T, T.size() * 4, hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(inv.data(), d_inv, inv.size() * 4, hipMemcpyDeviceToHost));
  FILE* fo = fopen(argv[3], "wb");
  if (!fo) { perror(argv[3]); return 1; }
  fwrite(T.data(), 4, T.size(), fo);
  fwrite(inv.data(), 4, inv.size(), fo);
  fclose(fo);
  printf("registered %d pairs of %d points, %d iterations; last transform of pair 0:\n", P, N, n_iter);
  for (int r = 0; r < 3; ++r) {
    const float* t = &T[(size_t)(n_iter - 1) * 12 + r * 4];
    printf("  %9.6f %9.6f %9.6f %9.6f\n", t[0], t[1], t[2], t[3]);
  }
  hipFree(d_src); hipFree(d_ref); hipFree(d_T); hipFree(d_inv);
  dsir_destroy(ctx);
  return 0;
}
