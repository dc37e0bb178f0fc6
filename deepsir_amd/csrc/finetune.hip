// Adam fine-tune of a registration result (the `use_tune` branch of the reference's
// pose_optimization): transformation_finetune (test.py:159-207) with HighDimSmoothL1Loss (test.py:103-131) and the 6-D
// rotation parametrisation Transformation / ortho2rotation (network/DGR.py:60-132).
//
//   params  rot6d = (a, b) in R^3 x R^3 (initialised with the first two columns of R), trans = t
//   R(a, b) x = a / |a|,  u = b - x (x.b) / (x.x),  y = u / |u|,  z = x cross y,  R = [x y z]   (norms clamped at 1e-8)
//   loss    L = sum_i w_i l_i / sum_i w_i,  s_i = |(R p_i + t - q_i) / quant|^2,
//           l_i = s_i / 2 (s_i < 1)  or  (sqrt(s_i + eps32) - 1/2) / 2
//   Adam    lr 0.1 * 0.999^k, betas (0.9, 0.999), eps 1e-8; stop: L < 1e-7, max_iter steps, or the relative change of L
//           below break_ratio for the max_break-th time (the counter is never reset, and the first step always counts:
//           the reference compares the first loss with itself)
//
// The reference runs this for one pair on the host side of torch (a Python loop of ~1000 dependent autograd steps).
// Here every pair of a batch is one 1024-thread workgroup that keeps the whole optimisation on its CU: each step is one
// pass over the pair's matched points (120 kB, L2-resident) that accumulates the loss and the 12 pose gradients
// dL/dR = sum g_i p_i^T, dL/dt = sum g_i (fp32 per point, fp64 across the block: two barriers), then one thread does the
// Gram-Schmidt backward pass and the Adam update in fp32 and publishes the new R, t through LDS.  No host round trip, no
// launch per step.  The branch is switched off in the reference and test.py cannot be imported here: parity unpinned,
// the rule is restated in oracle/finetune.py.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int FT_THREADS = 1024;
constexpr float kEps32 = 1.1920928955078125e-07f;   // np.finfo(np.float32).eps (test.py:104)

struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ Vec3 add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3 sub(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3 mul(Vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }

// ortho2rotation (DGR.py:60-108): R columns x, y, z from (a, b); also what the backward pass needs
struct Frame { Vec3 x, y, z, u; float na, nu, n2, s; };
__device__ __forceinline__ Frame make_frame(Vec3 a, Vec3 b) {
  Frame f;
  f.na = fmaxf(sqrtf(dot(a, a)), 1e-8f);
  f.x = mul(a, 1.f / f.na);
  f.n2 = fmaxf(dot(f.x, f.x), 1e-8f);
  f.s = dot(f.x, b) / f.n2;
  f.u = sub(b, mul(f.x, f.s));
  f.nu = fmaxf(sqrtf(dot(f.u, f.u)), 1e-8f);
  f.y = mul(f.u, 1.f / f.nu);
  f.z = cross(f.x, f.y);
  return f;
}

// gradient of the loss w.r.t. (a, b) from dL/dx, dL/dy, dL/dz (the columns of dL/dR)
__device__ __forceinline__ void frame_backward(const Frame& f, Vec3 b, Vec3 gx, Vec3 gy, Vec3 gz, Vec3& ga, Vec3& gb) {
  // z = x cross y
  Vec3 gxt = add(gx, cross(f.y, gz));
  Vec3 gyt = add(gy, cross(gz, f.x));
  // y = u / |u|   (|u| above the clamp: d y = (I - y y^T) / |u| du; at the clamp: du / 1e-8)
  const bool uc = sqrtf(dot(f.u, f.u)) < 1e-8f;
  Vec3 gu = uc ? mul(gyt, 1.f / f.nu) : mul(sub(gyt, mul(f.y, dot(f.y, gyt))), 1.f / f.nu);
  // u = b - x s, s = (x.b) / n2, n2 = x.x
  const float gux = dot(gu, f.x);
  gb = sub(gu, mul(f.x, gux / f.n2));
  const bool nc = dot(f.x, f.x) < 1e-8f;
  Vec3 ds_dx = nc ? mul(b, 1.f / f.n2) : sub(mul(b, 1.f / f.n2), mul(f.x, 2.f * f.s / f.n2));
  gxt = sub(gxt, add(mul(gu, f.s), mul(ds_dx, gux)));
  // x = a / |a|
  const bool ac = f.na <= 1e-8f;
  ga = ac ? mul(gxt, 1.f / f.na) : mul(sub(gxt, mul(f.x, dot(f.x, gxt))), 1.f / f.na);
}

struct FinetuneArgs {
  const float* src; const float* ref; const float* w;   // [pairs][m][3], [pairs][m][3], [pairs][m] (or nullptr: unweighted mean)
  const float* T_init;                                   // [pairs][3][4]
  int m, sigmoid, max_iter, max_break;
  float quant, break_ratio;
  float* T_out;                                          // [pairs][3][4]
  double* stats;                                         // [pairs][3]: iterations, loss, break_count (or nullptr)
};

__global__ __launch_bounds__(FT_THREADS) void pose_finetune_kernel(FinetuneArgs p) {
  __shared__ double red[16][13];
  __shared__ float s_R[9], s_t[3];
  __shared__ double s_W;
  __shared__ int s_stop;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* S = p.src + (int64_t)pair * p.m * 3;
  const float* Q = p.ref + (int64_t)pair * p.m * 3;
  const float* Wp = p.w ? p.w + (int64_t)pair * p.m : nullptr;
  auto weight = [&](int i) -> float {
    if (!Wp) return 1.f;
    const float v = Wp[i];
    return p.sigmoid ? 1.f / (1.f + expf(-v)) : v;
  };
  // W = sum of the weights (HighDimSmoothL1Loss.w1); the unweighted loss is the mean
  {
    double acc = 0.0;
    for (int i = tid; i < p.m; i += FT_THREADS) acc += (double)weight(i);
    acc = wave_sum(acc);
    if (lane == 0) red[wv][0] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int k = 0; k < FT_THREADS / 64; ++k) t += red[k][0];
      s_W = t;
    }
    __syncthreads();
  }
  // optimiser state (thread 0)
  Vec3 a, b, tr;
  float m1[9] = {}, m2[9] = {};
  double lr = 0.1;
  float loss_prev = 0.f, loss_last = 0.f;
  int brk = 0, it_last = -1;
  if (tid == 0) {
    const float* T = p.T_init + (int64_t)pair * 12;
    a = v3(T[0], T[4], T[8]);      // first column of R
    b = v3(T[1], T[5], T[9]);      // second column
    tr = v3(T[3], T[7], T[11]);
    const Frame f = make_frame(a, b);
    s_R[0] = f.x.x; s_R[1] = f.y.x; s_R[2] = f.z.x;
    s_R[3] = f.x.y; s_R[4] = f.y.y; s_R[5] = f.z.y;
    s_R[6] = f.x.z; s_R[7] = f.y.z; s_R[8] = f.z.z;
    s_t[0] = tr.x; s_t[1] = tr.y; s_t[2] = tr.z;
    s_stop = p.max_iter <= 0 ? 1 : 0;
  }
  __syncthreads();
  const float iq = 1.f / p.quant;
  for (int it = 0; it < p.max_iter; ++it) {
    if (s_stop) break;                                    // block-uniform (read after a barrier)
    const float R0 = s_R[0], R1 = s_R[1], R2 = s_R[2], R3 = s_R[3], R4 = s_R[4], R5 = s_R[5], R6 = s_R[6], R7 = s_R[7], R8 = s_R[8];
    const float t0 = s_t[0], t1 = s_t[1], t2 = s_t[2];
    float acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0.f;
    for (int i = tid; i < p.m; i += FT_THREADS) {
      const float px = S[3 * i], py = S[3 * i + 1], pz = S[3 * i + 2];
      const float ox = px * R0 + py * R1 + pz * R2 + t0;
      const float oy = px * R3 + py * R4 + pz * R5 + t1;
      const float oz = px * R6 + py * R7 + pz * R8 + t2;
      const float rx = (ox - Q[3 * i]) * iq, ry = (oy - Q[3 * i + 1]) * iq, rz = (oz - Q[3 * i + 2]) * iq;
      const float s = rx * rx + ry * ry + rz * rz;
      const float wi = weight(i);
      float l, c;                                         // loss term, d l / d s * 2 (so that d l / d r = c r)
      if (s < 1.f) { l = 0.5f * s; c = 1.f; }
      else { const float q = sqrtf(s + kEps32); l = 0.5f * (q - 0.5f); c = 0.5f / q; }
      acc[0] += wi * l;
      const float k = wi * c * iq;                        // d (w l) / d out = w c r / quant
      const float gx = k * rx, gy = k * ry, gz = k * rz;
      acc[1] += gx; acc[2] += gy; acc[3] += gz;
      acc[4] += gx * px; acc[5] += gx * py; acc[6] += gx * pz;   // dL/dR row 0
      acc[7] += gy * px; acc[8] += gy * py; acc[9] += gy * pz;
      acc[10] += gz * px; acc[11] += gz * py; acc[12] += gz * pz;
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const double v = wave_sum((double)acc[k]);
      if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
      double tot[13];
      for (int k = 0; k < 13; ++k) {
        double t = 0.0;
        for (int q = 0; q < FT_THREADS / 64; ++q) t += red[q][k];
        tot[k] = t;
      }
      const double Wn = Wp ? s_W : (double)p.m;
      const float loss = (float)(tot[0] / Wn);
      if (it == 0) loss_prev = loss;
      loss_last = loss; it_last = it;
      if (loss < 1e-7f) {
        s_stop = 1;
      } else {
        const float invW = (float)(1.0 / Wn);
        const Vec3 gt = v3((float)tot[1] * invW, (float)tot[2] * invW, (float)tot[3] * invW);
        // columns of dL/dR: dL/dx = (G00, G10, G20) ...
        const Vec3 gx = v3((float)tot[4] * invW, (float)tot[7] * invW, (float)tot[10] * invW);
        const Vec3 gy = v3((float)tot[5] * invW, (float)tot[8] * invW, (float)tot[11] * invW);
        const Vec3 gz = v3((float)tot[6] * invW, (float)tot[9] * invW, (float)tot[12] * invW);
        const Frame f = make_frame(a, b);
        Vec3 ga, gb;
        frame_backward(f, b, gx, gy, gz, ga, gb);
        const float g[9] = {ga.x, ga.y, ga.z, gb.x, gb.y, gb.z, gt.x, gt.y, gt.z};
        float prm[9] = {a.x, a.y, a.z, b.x, b.y, b.z, tr.x, tr.y, tr.z};
        // torch.optim.Adam, step it + 1, lr = 0.1 * 0.999^it
        const double bc1 = 1.0 - pow(0.9, (double)(it + 1)), bc2 = 1.0 - pow(0.999, (double)(it + 1));
        const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
        for (int k = 0; k < 9; ++k) {
          m1[k] = m1[k] + (g[k] - m1[k]) * 0.1f;
          m2[k] = m2[k] * 0.999f + (0.001f * g[k]) * g[k];
          const float denom = sqrtf(m2[k]) / bc2s + 1e-8f;
          prm[k] = prm[k] - step_size * (m1[k] / denom);
        }
        lr *= 0.999;
        a = v3(prm[0], prm[1], prm[2]); b = v3(prm[3], prm[4], prm[5]); tr = v3(prm[6], prm[7], prm[8]);
        const Frame fn = make_frame(a, b);
        s_R[0] = fn.x.x; s_R[1] = fn.y.x; s_R[2] = fn.z.x;
        s_R[3] = fn.x.y; s_R[4] = fn.y.y; s_R[5] = fn.z.y;
        s_R[6] = fn.x.z; s_R[7] = fn.y.z; s_R[8] = fn.z.z;
        s_t[0] = tr.x; s_t[1] = tr.y; s_t[2] = tr.z;
        if (fabsf(loss_prev - loss) < loss_prev * p.break_ratio) {
          ++brk;
          if (brk >= p.max_break) s_stop = 1;
        }
        loss_prev = loss;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    float* T = p.T_out + (int64_t)pair * 12;
    T[0] = s_R[0]; T[1] = s_R[1]; T[2] = s_R[2]; T[3] = s_t[0];
    T[4] = s_R[3]; T[5] = s_R[4]; T[6] = s_R[5]; T[7] = s_t[1];
    T[8] = s_R[6]; T[9] = s_R[7]; T[10] = s_R[8]; T[11] = s_t[2];
    if (p.stats) {
      double* st = p.stats + (int64_t)pair * 3;
      st[0] = (double)it_last; st[1] = (double)loss_last; st[2] = (double)brk;
    }
  }
}

}  // namespace

void launch_pose_finetune(const float* src, const float* ref, const float* w, int sigmoid, int pairs, int m, const float* T_init,
                          float quant, int max_iter, float break_ratio, int max_break, float* T_out, double* stats,
                          hipStream_t st) {
  FinetuneArgs a;
  a.src = src; a.ref = ref; a.w = w; a.T_init = T_init; a.m = m; a.sigmoid = sigmoid; a.max_iter = max_iter; a.max_break = max_break;
  a.quant = quant; a.break_ratio = break_ratio; a.T_out = T_out; a.stats = stats;
  hipLaunchKernelGGL(pose_finetune_kernel, dim3(pairs), dim3(FT_THREADS), 0, st, a);
}

}  // namespace dsir
