// Host side of libdsir.so: context, weights, workspace, the launch schedules of
// RandLA.forward / aggregation / forward_align_4, and the C ABI of include/dsir.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "dsir.h"
#include "kernels.h"

using namespace dsir;

// ------------------------------------------------------------------ the tuning gate
// The ONLY place of the library that reads the environment.  Every DSIR_* measurement / A-B switch goes through
// tuning_env(); unless the gate is open - DSIR_TUNING=1 in the environment, or dsir_set_tuning(1) before the first call
// that reads a switch (most are read once into function-static state) - the library ignores every DSIR_* variable, so a
// stray one in a user's environment cannot change kernel selection.
static int g_tuning = -1;     // -1: not decided yet, 0: closed, 1: open
const char* dsir::tuning_env(const char* name) {
  if (g_tuning < 0) {
    const char* e = getenv("DSIR_TUNING");
    g_tuning = (e && e[0] == '1' && e[1] == 0) ? 1 : 0;
  }
  return g_tuning == 1 ? getenv(name) : nullptr;
}

namespace {

thread_local std::string g_create_error;

// ------------------------------------------------------------------ parameters
struct HostParam {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
  bool ignored = false;  // num_batches_tracked
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Mlp2dW { const float *W = nullptr, *b = nullptr, *gamma = nullptr, *beta = nullptr; int cin = 0, cout = 0, groups = 0; };
struct AttW { const float* fc = nullptr; const float* fc_g = nullptr; int d = 0; Mlp2dW mlp; };   // fc_g: see up_fc_g
struct BlockW { Mlp2dW mlp1, lfa1, lfa2, mlp2, skip; AttW att1, att2; int d_in = 0, d = 0; const float* lse_w8 = nullptr;   // lse_w8: up_lse_uv
                const float *pair_W = nullptr, *pair_b = nullptr; };   // mlp1's rows followed by mlp_skip's (and the biases likewise): up_pair
struct LinW { const float *W = nullptr, *b = nullptr; int cin = 0, cout = 0; };
struct RandlaW { Mlp2dW pre; BlockW blk[4]; Mlp2dW mid; Mlp2dW dec[4]; const float* out_w = nullptr; int dec_out = 0; LinW fc[3]; int cin = 0, ncls = 0;
                 const void* head_wh[4] = {}; const void* head_wl[4] = {}; };   // fp16 split of mlp_out + fc_label (head_mlp_h.hip)
struct NetW { RandlaW feat, inl; LinW mlp_feat[3], mlp_att[5], mlp_proj; };

// ------------------------------------------------------------------ workspace
struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0;
  bool overflow = false;
  void* raw(size_t bytes) {
    size_t a = (top + 255) & ~(size_t)255;
    if (a + bytes > cap) { overflow = true; return base; }
    top = a + bytes;
    return base + a;
  }
  template <typename T> T* get(size_t count) { return reinterpret_cast<T*>(raw(count * sizeof(T))); }
  size_t mark() const { return top; }
  void release(size_t m) { top = m; }
};

struct Pyramid {     // KNN pyramid of a cloud batch, levels concatenated (data_base.py:178-181)
  int clouds = 0, n = 0;
  int nl[DSIR_MAX_LEVELS + 1] = {};
  int off[DSIR_MAX_LEVELS + 1] = {};   // level offsets into xyz / neigh / interp
  int soff[DSIR_MAX_LEVELS + 1] = {};  // level offsets into sub
  int S = 0, S1 = 0;
  const float* xyz = nullptr;     // [clouds][S][3]
  const int32_t* neigh = nullptr; // [clouds][S][16]
  const int32_t* sub = nullptr;   // [clouds][S1][16]
  const int32_t* interp = nullptr;// [clouds][S]
};

// a tensor with a lazily applied GroupNorm (+activation)
struct Act {
  float* p = nullptr;
  int C = 0;
  int rows = 0;       // rows per cloud
  GnRef gn = {nullptr, nullptr, nullptr, 0, 0.0};
  int act = 0;
  // lse_uv.hip: the position-encoding layer of levels 0 / 1 is not in memory (p == nullptr): per-point tables instead
  const float* uv = nullptr;      // [clouds][rows / 16][2 C]
  const float* dist = nullptr;    // [clouds][rows]
  const float* w8 = nullptr;
};

}  // namespace

struct dsir_ctx {
  int device = 0;
  dsir_cfg cfg{};
  hipStream_t stream = nullptr;        // where every launch of this context goes: own_stream, or a caller's (dsir_set_stream)
  hipStream_t own_stream = nullptr;
  std::string err;
  const char* sched_error = nullptr;     // a launcher refused a layer (outside its envelope): reported by the schedule's caller
  std::vector<HostParam> params;
  std::unordered_map<std::string, int> index;
  float* dweights = nullptr;
  uint16_t* dweights16 = nullptr;        // fp16 split of the WHOLE weight blob: high parts [0, n), low parts [n, 2 n), same offsets
  size_t nweights = 0;                   // floats in dweights
  const void* agg_wh[5] = {}; const void* agg_wl[5] = {};
  bool finalized = false;
  NetW net;
  Arena ws;
  double* stats = nullptr;   // GroupNorm statistics slots
  size_t stats_cap = 0, stats_top = 0;
  size_t stats_base = 0;          // dsir_register: the passes of one call take consecutive regions of an arena zeroed ONCE
  bool stats_prezeroed = false;
  // nn_match timing
  bool time_match = false;
  // hipGraph replay of dsir_register (launch-bound small batches)
  bool use_graph = false;
  // captured registrations, one per distinct call signature (sizes AND buffer addresses): a server that batches 1 .. K
  // single-pair requests into one call replays K graphs in turn (deepsir_amd/serve.py); the oldest is evicted beyond kMaxGraphs
  struct Graph { std::vector<unsigned char> key; hipGraphExec_t exec; void* walk_block; };   // walk_block: the graph's walker programs (device)
  int64_t graph_nodes[4] = {0, 0, 0, 0};   // the latest captured registration: nodes in all, kernel / memset / memcpy nodes (dsir_graph_stats)
  // ---- independent branches of the schedule on auxiliary streams (fork / join through events; captured into a registration's graph as
  // parallel branches).  With a few clouds in flight the chip is nearly empty and a registration is one long chain of dependent
  // launches; the KNN searches of the four levels, a level's position-encoding branch (lfa.mlp1 -> lfa.mlp2), its mlp_skip and the
  // loop-invariant halves of the aggregation do not depend on the chain beside them and can run beside it.  Same kernels, same
  // operands: same bits (tests/test_gpu_walk.py).  An experiment that did NOT pay (see fork_mode): kept as a switch.
  static constexpr int kAux = 2;
  static constexpr int kForkClouds = 16;
  hipStream_t aux[kAux] = {nullptr, nullptr};
  std::vector<hipEvent_t> fork_events;
  size_t fork_events_used = 0;
  int fork_mode = 0;                           // 1: fork (dsir_enable_fork / DSIR_FORK=1).  OFF by default: measured SLOWER - one pair replayed
                                               // from its graph 3.08 ms on one stream, 3.37 - 4.06 ms with any of the branches forked (a
                                               // captured graph with parallel branches leaves the runtime's single-queue fast path:
                                               // even ONE fork / join costs 0.3 ms; profiles/README.md round 5)
  // ---- deep-level walker (walk.hip): the programs of one call's RandLA passes live in device memory
  static constexpr int kWalkSlots = 12;        // programs per call (1 extractor pass or 2, up to 10 inlier passes)
  static constexpr int kWalkClouds = 16;       // the walker serves launches of up to that many clouds
  int walk_mode = 0;                           // 1: the deep levels of a pass as one launch (dsir_enable_walk / DSIR_WALK=1); OFF by default -
                                               // measured slower than the launches it replaces (walk.hip, "What it measured")
  int walk_used = 0;                           // programs of the current call
  WalkProgram* walk_dev = nullptr;             // eager calls: device programs, filled by in-stream copies from ...
  WalkProgram* walk_host[2] = {nullptr, nullptr};   // ... pinned staging, two sets taken in turn by consecutive calls
  hipEvent_t walk_ev[2] = {nullptr, nullptr};  // recorded after a call's last copy from the set
  bool walk_ev_armed[2] = {false, false};
  int walk_set = 0;
  unsigned* walk_ctr = nullptr;                // [kWalkSlots][kWalkClouds][kWalkCtrWords] tile queues / completion counters
  unsigned long long* walk_trace = nullptr;    // measurement (DSIR_WALK_TRACE, dsir_walk_trace): [kWalkSlots][kWalkMaxPhases][4] device-clock stamps
  int walk_wpc = 0;                            // tuning hook (DSIR_WALK_WPC): workgroups per cloud, 0 = by launch size
  int walk_flags = 0;                          // tuning hook (DSIR_WALK_FLAGS): WalkProgram::flags
  // a registration under capture: programs are collected on the host and uploaded ONCE, after the capture, into the graph's own block
  bool capturing = false;
  std::vector<unsigned char> cap_host;
  WalkProgram* cap_dev = nullptr;
  std::vector<Graph> graphs;
  static constexpr size_t kMaxGraphs = 16;
  void drop_graphs() {
    for (auto& g : graphs) { hipGraphExecDestroy(g.exec); if (g.walk_block) hipFree(g.walk_block); }
    graphs.clear();
  }
  struct MatchEvents { hipEvent_t op0, op1, k0, k1; };   // whole operation / its dominant kernel alone
  std::vector<MatchEvents> match_events;
  size_t match_events_used = 0;
  double match_ms = 0.0, match_kernel_ms = 0.0;
  int64_t match_launches = 0;
  // arg-min path of dsir_register: 1 = screened (nn_screen.hip) for large problems, 0 = always the exhaustive kernel
  int screen_mode = 1;
  int prune_min_points = 8192;          // pruned search (nn_prune.hip) for ref clouds of that many points and more; 0 = off
  long long prune_min_rows = 65536;     // ... in launches of that many src rows (pairs x points) and more
  // aggregation chain: 1 = fp16-split products on the fp16 matrix pipe (agg_chain_h.hip), 0 = exact-fp32 chain (agg_chain.hip)
  int agg_split = 1;
  int kabsch_chunked_min = 0;           // clouds of that many points and more solve their pose in chunks; 0 = kKabschChunkedMin
  // device-clock brackets {first wave start, last wave end} of the timed nn_match launches
  unsigned long long* match_ts = nullptr;    // [kMatchSlots][2]
  size_t match_ts_used = 0;
  double match_dev_ms = 0.0;
  int64_t match_dev_launches = 0;
  // running totals of the screened arg-min inside dsir_register (dsir_screen_stats)
  unsigned long long* screen_acc = nullptr;   // device, 4 x u64 (+ 2 x u64: tile products kept / in all by the pruned search)
  int64_t exhaustive_searches = 0;            // searches that took the exhaustive kernel directly (small problems)
};
constexpr size_t kMatchSlots = 4096;

namespace {

int fail(dsir_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return 1;
}

#define HIP_OK(c, expr)                                                                 \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) return fail((c), "%s: %s", #expr, hipGetErrorString(e__));   \
  } while (0)

static_assert(kMaxLevels == DSIR_MAX_LEVELS, "kernels.h and dsir.h disagree on the level count");

void level_sizes(const dsir_cfg& cfg, int n, int* nl) {
  nl[0] = n;
  for (int l = 0; l < cfg.num_layers; ++l) nl[l + 1] = nl[l] / cfg.sub_sampling_ratio[l];
}

void fill_pyramid_layout(const dsir_cfg& cfg, int clouds, int n, Pyramid& p) {
  p.clouds = clouds; p.n = n;
  level_sizes(cfg, n, p.nl);
  p.off[0] = 0; p.soff[0] = 0;
  for (int l = 0; l < cfg.num_layers; ++l) { p.off[l + 1] = p.off[l] + p.nl[l]; p.soff[l + 1] = p.soff[l] + p.nl[l + 1]; }
  p.S = p.off[cfg.num_layers]; p.S1 = p.soff[cfg.num_layers];
}

// The most contributions any (cloud, group) GroupNorm statistic receives when a cloud of n points goes through RandLA.forward: the
// maximum of the launchers' own counts (kernels.h) over every MLP2D of the schedule, under the default dispatch of launch_pw_gemm
// (Cin <= 64 and the relative-position layers: pw_stream.hip or - d / 2 = 8, 32 - lse_uv.hip; wider: pw_tile.hip, whose count also
// bounds the generic pw_gemm.hip kernel's 64-row blocks).  The exactness proof of the statistics' atomics (device_utils.h) needs
// this number <= kGnMaxContrib: dsir_create refuses a max_points beyond it.
int gn_max_contributions(const dsir_cfg& g, int n) {
  int worst = 0;
  auto layer = [&](int M, int cin, int cout) {
    const int groups = cout >= 64 ? 8 : 4;
    const int c = cin <= 64 ? pw_stream_gn_contributions(M, cout) : pw_tile_gn_contributions(M, cout, groups);
    if (c > worst) worst = c;
  };
  int nl[DSIR_MAX_LEVELS + 1];
  level_sizes(g, n, nl);
  const int L = g.num_layers;
  layer(nl[0], 6 > g.feat_len ? 6 : g.feat_len, 8);
  int dim = 8;
  for (int l = 0; l < L; ++l) {
    const int d = g.d_out[l], m = nl[l], mk = nl[l] * kKnn;
    layer(m, dim, d / 2); layer(m, dim, 2 * d);                  // mlp1, mlp_skip
    if (d / 2 == 8 || d / 2 == 32) { const int c = lse_uv_gn_contributions(m, d / 2); if (c > worst) worst = c; }
    layer(mk, 10, d / 2);                                        // lfa.mlp1 (also when the tables are switched off)
    layer(mk, d / 2, d / 2);                                     // lfa.mlp2
    layer(m, d, d / 2); layer(m, d, d); layer(m, d, 2 * d);      // att_pooling_1.mlp, att_pooling_2.mlp, mlp2
    dim = 2 * d;
  }
  layer(nl[L], dim, dim);
  int dcur = dim;
  for (int j = 0; j < L; ++j) {
    const int lvl = L - 1 - j;
    const int cin = j < L - 1 ? dcur + 2 * g.d_out[L - j - 2] : 4 * g.d_out[0];
    dcur = j < L - 1 ? 2 * g.d_out[L - j - 2] : 2 * g.d_out[0];
    layer(nl[lvl], cin, dcur);
  }
  return worst;
}

// ------------------------------------------------------------------ expected state-dict (mirrors deepsir_amd/arch.py)
void add_param(dsir_ctx* c, const std::string& name, std::vector<int64_t> shape, bool ignored = false) {
  HostParam p;
  p.name = name; p.shape = std::move(shape); p.ignored = ignored;
  c->index[name] = (int)c->params.size();
  c->params.push_back(std::move(p));
}
void add_mlp2d(dsir_ctx* c, const std::string& pre, int cin, int cout) {
  add_param(c, pre + ".conv.weight", {cout, cin, 1, 1});
  add_param(c, pre + ".conv.bias", {cout});
  add_param(c, pre + ".norm.weight", {cout});
  add_param(c, pre + ".norm.bias", {cout});
}
void add_att(dsir_ctx* c, const std::string& pre, int din, int dout) {
  add_param(c, pre + ".fc.weight", {din, din, 1, 1});
  add_mlp2d(c, pre + ".mlp", din, dout);
}
void add_mlp1d(dsir_ctx* c, const std::string& pre, const std::vector<int>& ch) {
  int pos = 0;
  const int n = (int)ch.size();
  for (int i = 1; i < n; ++i) {
    const std::string p = pre + "." + std::to_string(pos);
    add_param(c, p + ".weight", {ch[i], ch[i - 1], 1});
    add_param(c, p + ".bias", {ch[i]});
    ++pos;
    if (i < n - 1) {
      const std::string q = pre + "." + std::to_string(pos);
      add_param(c, q + ".weight", {ch[i]});
      add_param(c, q + ".bias", {ch[i]});
      add_param(c, q + ".running_mean", {ch[i]});
      add_param(c, q + ".running_var", {ch[i]});
      add_param(c, q + ".num_batches_tracked", {}, true);
      pos += 2;
    }
  }
}
void add_randla(dsir_ctx* c, const std::string& pre, int cin, int ncls) {
  const dsir_cfg& g = c->cfg;
  int dim = 8;
  add_mlp2d(c, pre + ".mlp_pre", cin, dim);
  for (int i = 0; i < g.num_layers; ++i) {
    const int d = g.d_out[i];
    const std::string p = pre + ".dilated_res_blocks." + std::to_string(i);
    add_mlp2d(c, p + ".mlp1", dim, d / 2);
    add_mlp2d(c, p + ".lfa.mlp1", 10, d / 2);
    add_att(c, p + ".lfa.att_pooling_1", d, d / 2);
    add_mlp2d(c, p + ".lfa.mlp2", d / 2, d / 2);
    add_att(c, p + ".lfa.att_pooling_2", d, d);
    add_mlp2d(c, p + ".mlp2", d, 2 * d);
    add_mlp2d(c, p + ".mlp_skip", dim, 2 * d);
    dim = 2 * d;
  }
  add_mlp2d(c, pre + ".mlp_mid", dim, dim);
  int dcur = dim;
  const int L = g.num_layers;
  for (int j = 0; j < L; ++j) {
    int cin_j;
    if (j < L - 1) { cin_j = dcur + 2 * g.d_out[L - j - 2]; dcur = 2 * g.d_out[L - j - 2]; }
    else { cin_j = 4 * g.d_out[0]; dcur = 2 * g.d_out[0]; }
    add_mlp2d(c, pre + ".decoder_blocks." + std::to_string(j), cin_j, dcur);
  }
  add_param(c, pre + ".mlp_out.weight", {g.out_feat_dim, dcur, 1, 1});
  add_mlp1d(c, pre + ".fc_label", {g.out_feat_dim, 64, 32, ncls});
}

// ------------------------------------------------------------------ weight upload
struct Uploader {
  std::vector<float> blob;
  size_t put(const std::vector<float>& v) {
    size_t o = (blob.size() + 63) & ~(size_t)63;
    blob.resize(o + v.size());
    std::memcpy(blob.data() + o, v.data(), v.size() * sizeof(float));
    return o;
  }
};

const HostParam& P(dsir_ctx* c, const std::string& name) { return c->params[c->index.at(name)]; }

struct Mlp2dOff { size_t W, b, g, be; int cin, cout; };
Mlp2dOff up_mlp2d(dsir_ctx* c, Uploader& u, const std::string& pre) {
  const HostParam& w = P(c, pre + ".conv.weight");
  return {u.put(w.data), u.put(P(c, pre + ".conv.bias").data), u.put(P(c, pre + ".norm.weight").data),
          u.put(P(c, pre + ".norm.bias").data), (int)w.shape[1], (int)w.shape[0]};
}
Mlp2dW bind_mlp2d(const float* base, const Mlp2dOff& o) {
  Mlp2dW m;
  m.W = base + o.W; m.b = base + o.b; m.gamma = base + o.g; m.beta = base + o.be;
  m.cin = o.cin; m.cout = o.cout; m.groups = o.cout >= 64 ? 8 : 4;   // RandLANet.py:93
  return m;
}

struct LinOff { size_t W, b; int cin, cout; };
// Conv1d followed (optionally) by eval-mode BatchNorm1d, folded in double precision:
// y = ((W x + b) - mu) / sqrt(var + 1e-5) * g + beta    (RandLANet.py:39-43)
LinOff up_lin(dsir_ctx* c, Uploader& u, const std::string& pre, int pos, bool bn) {
  const HostParam& w = P(c, pre + "." + std::to_string(pos) + ".weight");
  const HostParam& b = P(c, pre + "." + std::to_string(pos) + ".bias");
  const int cout = (int)w.shape[0], cin = (int)w.shape[1];
  std::vector<float> W(w.data), B(b.data);
  if (bn) {
    const std::string q = pre + "." + std::to_string(pos + 1);
    const auto& g = P(c, q + ".weight").data; const auto& be = P(c, q + ".bias").data;
    const auto& mu = P(c, q + ".running_mean").data; const auto& var = P(c, q + ".running_var").data;
    for (int o = 0; o < cout; ++o) {
      const double s = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
      for (int i = 0; i < cin; ++i) W[(size_t)o * cin + i] = (float)((double)w.data[(size_t)o * cin + i] * s);
      B[o] = (float)(((double)b.data[o] - (double)mu[o]) * s + (double)be[o]);
    }
  }
  return {u.put(W), u.put(B), cin, cout};
}
LinW bind_lin(const float* base, const LinOff& o) { LinW l; l.W = base + o.W; l.b = base + o.b; l.cin = o.cin; l.cout = o.cout; return l; }

// Split attentive pooling (d >= 64): G = W1 f is consumed only by the pooling kernel, where lane (fr, fq) of a block needs
// the gathered G values of its column in each of the block's four 16-column tiles.  W1 = fc[:, :d/2] is therefore uploaded
// a second time with its rows permuted so that those four values are adjacent: G' column 64 b + 4 fr + t = the column of
// tile t, lane fr of block b - one 16-byte gather instead of four 4-byte ones.  Which columns form tile t of block b is
// the consumer's mapping: pw_stream.hip (d = 64, 128) pairs 32 columns of the gathered half with the matching 32 of the
// enc half, pw_tile.hip (d = 256) takes 64 consecutive columns.
size_t up_fc_g(Uploader& u, const HostParam& fc, int d) {
  if (d < 64) return 0;
  const int h = d / 2;
  std::vector<float> w((size_t)d * h);
  for (int b = 0; b < d / 64; ++b)
    for (int fr = 0; fr < 16; ++fr)
      for (int t = 0; t < 4; ++t) {
        const int pos = 64 * b + 4 * fr + t;
        const int src = d <= 128 ? (t < 2 ? 32 * b + 16 * t + fr : h + 32 * b + 16 * (t - 2) + fr) : 64 * b + 16 * t + fr;
        for (int k = 0; k < h; ++k) w[(size_t)pos * h + k] = fc.data[(size_t)src * d + k];
      }
  return u.put(w);
}

// lse_uv.hip: lfa.mlp1 of a level with d / 2 <= 32 channels, folded for the split by linearity
//   enc_raw[i, k][c] = a[c] dist + U[j][c] + V[i][c]:   per channel {a, ux, uy, uz, vx, vy, vz, b} with u = W[:, 1:4] + W[:, 7:10] (the
// neighbour's coordinates enter through the offset and through their own channels), v = W[:, 4:7] - W[:, 1:4], b = bias.
size_t up_lse_uv(Uploader& u, const HostParam& w, const HostParam& b, int kh) {
  if (kh != 8 && kh != 32) return 0;
  std::vector<float> f((size_t)kh * 8);
  for (int c = 0; c < kh; ++c) {
    const float* r = &w.data[(size_t)c * 10];
    f[c * 8 + 0] = r[0];
    for (int k = 0; k < 3; ++k) { f[c * 8 + 1 + k] = r[1 + k] + r[7 + k]; f[c * 8 + 4 + k] = r[4 + k] - r[1 + k]; }
    f[c * 8 + 7] = b.data[c];
  }
  return u.put(f);
}

struct RandlaOff {
  Mlp2dOff pre, mid, dec[4];
  struct { Mlp2dOff mlp1, lfa1, lfa2, mlp2, skip, a1m, a2m; size_t fc1, fc2, fc1g, fc2g, lse8, pair_w, pair_b; bool pair; } blk[4];
  size_t out_w; int dec_out;
  LinOff fc[3];
};
RandlaOff up_randla(dsir_ctx* c, Uploader& u, const std::string& pre) {
  RandlaOff r;
  r.pre = up_mlp2d(c, u, pre + ".mlp_pre");
  for (int i = 0; i < 4; ++i) {
    const std::string p = pre + ".dilated_res_blocks." + std::to_string(i);
    r.blk[i].mlp1 = up_mlp2d(c, u, p + ".mlp1");
    r.blk[i].lfa1 = up_mlp2d(c, u, p + ".lfa.mlp1");
    r.blk[i].lse8 = up_lse_uv(u, P(c, p + ".lfa.mlp1.conv.weight"), P(c, p + ".lfa.mlp1.conv.bias"), c->cfg.d_out[i] / 2);
    r.blk[i].fc1 = u.put(P(c, p + ".lfa.att_pooling_1.fc.weight").data);
    r.blk[i].fc1g = up_fc_g(u, P(c, p + ".lfa.att_pooling_1.fc.weight"), c->cfg.d_out[i]);
    r.blk[i].a1m = up_mlp2d(c, u, p + ".lfa.att_pooling_1.mlp");
    r.blk[i].lfa2 = up_mlp2d(c, u, p + ".lfa.mlp2");
    r.blk[i].fc2 = u.put(P(c, p + ".lfa.att_pooling_2.fc.weight").data);
    r.blk[i].fc2g = up_fc_g(u, P(c, p + ".lfa.att_pooling_2.fc.weight"), c->cfg.d_out[i]);
    r.blk[i].a2m = up_mlp2d(c, u, p + ".lfa.att_pooling_2.mlp");
    r.blk[i].mlp2 = up_mlp2d(c, u, p + ".mlp2");
    r.blk[i].skip = up_mlp2d(c, u, p + ".mlp_skip");
    // mlp1 and mlp_skip read the same input (RandLANet.py:226 / :229): where mlp1's width is a whole number of 64-column tiles the two
    // weight matrices are uploaded once more, one after the other, for a launch that computes both (GemmArgs::c_split)
    r.blk[i].pair = false; r.blk[i].pair_w = r.blk[i].pair_b = 0;
    {
      const HostParam& w1 = P(c, p + ".mlp1.conv.weight");
      const HostParam& w2 = P(c, p + ".mlp_skip.conv.weight");
      if (w1.shape[0] % 64 == 0 && w1.shape[1] == w2.shape[1]) {
        std::vector<float> w(w1.data), b(P(c, p + ".mlp1.conv.bias").data);
        w.insert(w.end(), w2.data.begin(), w2.data.end());
        const auto& b2 = P(c, p + ".mlp_skip.conv.bias").data;
        b.insert(b.end(), b2.begin(), b2.end());
        r.blk[i].pair_w = u.put(w); r.blk[i].pair_b = u.put(b); r.blk[i].pair = true;
      }
    }
  }
  r.mid = up_mlp2d(c, u, pre + ".mlp_mid");
  for (int j = 0; j < 4; ++j) r.dec[j] = up_mlp2d(c, u, pre + ".decoder_blocks." + std::to_string(j));
  const HostParam& ow = P(c, pre + ".mlp_out.weight");
  r.out_w = u.put(ow.data); r.dec_out = (int)ow.shape[1];
  r.fc[0] = up_lin(c, u, pre + ".fc_label", 0, true);
  r.fc[1] = up_lin(c, u, pre + ".fc_label", 3, true);
  r.fc[2] = up_lin(c, u, pre + ".fc_label", 6, false);
  return r;
}
RandlaW bind_randla(const float* base, const RandlaOff& o, const dsir_cfg& g) {
  RandlaW r;
  r.pre = bind_mlp2d(base, o.pre);
  r.cin = o.pre.cin;
  for (int i = 0; i < 4; ++i) {
    BlockW& b = r.blk[i];
    b.mlp1 = bind_mlp2d(base, o.blk[i].mlp1); b.lfa1 = bind_mlp2d(base, o.blk[i].lfa1);
    b.lfa2 = bind_mlp2d(base, o.blk[i].lfa2); b.mlp2 = bind_mlp2d(base, o.blk[i].mlp2);
    b.skip = bind_mlp2d(base, o.blk[i].skip);
    b.att1.fc = base + o.blk[i].fc1; b.att1.d = g.d_out[i]; b.att1.mlp = bind_mlp2d(base, o.blk[i].a1m);
    b.att2.fc = base + o.blk[i].fc2; b.att2.d = g.d_out[i]; b.att2.mlp = bind_mlp2d(base, o.blk[i].a2m);
    b.att1.fc_g = g.d_out[i] >= 64 ? base + o.blk[i].fc1g : nullptr;
    b.att2.fc_g = g.d_out[i] >= 64 ? base + o.blk[i].fc2g : nullptr;
    b.lse_w8 = (g.d_out[i] == 16 || g.d_out[i] == 64) ? base + o.blk[i].lse8 : nullptr;
    b.d = g.d_out[i]; b.d_in = b.mlp1.cin;
    if (o.blk[i].pair) { b.pair_W = base + o.blk[i].pair_w; b.pair_b = base + o.blk[i].pair_b; }
  }
  r.mid = bind_mlp2d(base, o.mid);
  for (int j = 0; j < 4; ++j) r.dec[j] = bind_mlp2d(base, o.dec[j]);
  r.out_w = base + o.out_w; r.dec_out = o.dec_out;
  for (int k = 0; k < 3; ++k) r.fc[k] = bind_lin(base, o.fc[k]);
  r.ncls = r.fc[2].cout;
  return r;
}

// A/B switch: DSIR_NO_ATT_POOL = the round-3 EPI_ATT / EPI_ATT2 kernels (pw_stream.hip) for d = 16 / 64 / 128 instead of att_pool.hip
bool att_pool_enabled() {
  static const bool off = tuning_flag("DSIR_NO_ATT_POOL");
  return !off;
}

// A/B switch: DSIR_NO_LSE_UV = lfa.mlp1 of levels 0 / 1 written to memory as up to round 3 (pw_stream.hip, loader S_LSE) instead of
// the per-point tables of lse_uv.hip; the tables' consumers are att_pool.hip and pw_stream.hip (loader S_UV) only
bool lse_uv_enabled() {
  static const bool off = tuning_flag("DSIR_NO_LSE_UV") || tuning_flag("DSIR_NO_STREAM");   // the tables' GEMM consumer is pw_stream.hip alone
  return !off && att_pool_enabled();
}

// ------------------------------------------------------------------ schedule helpers
// Upload a finished walker program and launch it (walk.hip).  Eager calls: in-stream copy from pinned staging; a registration under
// capture: collected on the host, uploaded once after the capture (dsir_register), the kernel node holds the final device address.
int walk_flush(dsir_ctx* c, WalkProgram& P, hipStream_t st) {
  if (P.nphases <= 0) return 0;
  if (c->walk_used >= dsir_ctx::kWalkSlots) return fail(c, "walker: more than %d programs in one call", dsir_ctx::kWalkSlots);
  const int slot = c->walk_used++;
  P.ctr = c->walk_ctr + (size_t)slot * dsir_ctx::kWalkClouds * kWalkCtrWords;
  P.trace = c->walk_trace ? c->walk_trace + (size_t)slot * kWalkMaxPhases * 4 : nullptr;
  const size_t bytes = offsetof(WalkProgram, job) + (size_t)P.nphases * sizeof(WalkJob);
  const WalkProgram* dev;
  if (c->capturing) {
    std::memcpy(c->cap_host.data() + (size_t)slot * sizeof(WalkProgram), &P, bytes);
    dev = c->cap_dev + slot;
  } else {
    WalkProgram* h = c->walk_host[c->walk_set] + slot;
    std::memcpy(reinterpret_cast<void*>(h), &P, bytes);
    HIP_OK(c, hipMemcpyAsync(c->walk_dev + slot, h, bytes, hipMemcpyHostToDevice, st));
    dev = c->walk_dev + slot;
  }
  // queues and counters of this program: part of the region a registration zeroes in its opening launch; otherwise here
  if (!c->stats_prezeroed) HIP_OK(c, hipMemsetAsync(P.ctr, 0, sizeof(unsigned) * P.clouds * kWalkCtrWords, st));
  launch_walk(P, dev, st);
  P.nphases = 0;
  return 0;
}
// a call that may run RandLA passes: its programs start at slot 0; eager calls take the other staging set (the previous call's copies
// may still be queued) after making sure that set's own last copy has run
int walk_begin_call(dsir_ctx* c) {
  c->walk_used = 0;
  c->fork_events_used = 0;
  if (c->capturing || !c->walk_dev) return 0;
  c->walk_set ^= 1;
  if (c->walk_ev_armed[c->walk_set]) { HIP_OK(c, hipEventSynchronize(c->walk_ev[c->walk_set])); c->walk_ev_armed[c->walk_set] = false; }
  return 0;
}
int walk_end_call(dsir_ctx* c) {
  if (c->capturing || !c->walk_dev || c->walk_used == 0) return 0;
  HIP_OK(c, hipEventRecord(c->walk_ev[c->walk_set], c->stream));
  c->walk_ev_armed[c->walk_set] = true;
  return 0;
}

// `to` continues after everything enqueued on `from` so far (fork: main -> auxiliary; join: auxiliary -> main).  Events are taken
// from a per-context pool that restarts with every call; a wait binds to the event's latest record at the time it is enqueued.
hipEvent_t fork_mark(dsir_ctx* c, hipStream_t from) {
  if (c->fork_events_used == c->fork_events.size()) {
    hipEvent_t e = nullptr;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    c->fork_events.push_back(e);
  }
  hipEvent_t e = c->fork_events[c->fork_events_used++];
  hipEventRecord(e, from);
  return e;
}
void fork_wait(hipStream_t to, hipEvent_t e) { hipStreamWaitEvent(to, e, 0); }
void fork_to(dsir_ctx* c, hipStream_t from, hipStream_t to) { fork_wait(to, fork_mark(c, from)); }
// what: 1 = the KNN pyramid's levels, 2 = a block's position-encoding branch, 4 = a block's mlp_skip, 8 = the aggregation's loop invariants
// (DSIR_FORK_MASK: tuning hook - a dependency between streams has a price of its own, only the longer branches pay for it)
// The auxiliary streams of the forked schedule exist only in a context that forks: every stream a process creates takes a hardware
// queue in turn (GPU_MAX_HW_QUEUES of them), and two engines' main streams that land on one queue do not overlap - creating two idle
// streams per context moved the serving scheduler's second engine onto the first one's queue (profiles/r05_serving_queues.txt).
void ensure_aux_streams(dsir_ctx* c) {
  if (c->aux[0]) return;
  for (int k = 0; k < dsir_ctx::kAux; ++k)
    if (hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking) != hipSuccess) c->aux[k] = nullptr;
  if (!c->aux[0] || !c->aux[1]) { for (int k = 0; k < dsir_ctx::kAux; ++k) { if (c->aux[k]) hipStreamDestroy(c->aux[k]); c->aux[k] = nullptr; } }
}

bool fork_on(const dsir_ctx* c, int clouds, int what) {
  static const int mask = (int)tuning_int("DSIR_FORK_MASK", 15);
  return c->fork_mode && (mask & what) && c->aux[0] && clouds <= dsir_ctx::kForkClouds;
}

struct Sched {
  dsir_ctx* c;
  hipStream_t st;
  int clouds;
  // deep-level walker: while `rec` is set, layers whose kernel the walker holds become PHASES of one launch instead of launches
  WalkProgram* rec = nullptr;
  int rec_wpc = 1;
  int rec_error = 0;
  // launch what has been recorded; recording stops when the call has no program slot left (the rest of the pass: plain launches)
  void rec_flush() {
    if (!rec) return;
    if (rec->nphases > 0 && walk_flush(c, *rec, st)) rec_error = 1;
    if (c->walk_used >= dsir_ctx::kWalkSlots) rec = nullptr;
  }
  // false: not recorded (no room) - the caller launches the layer itself
  bool rec_push(const WalkJob& j) {
    if (!rec) return false;
    if (rec->nphases == kWalkMaxPhases) { rec_flush(); if (!rec) return false; }
    WalkJob& d = rec->job[rec->nphases];
    d = j;
    d.dep = rec->nphases > 0 ? rec->nphases - 1 : -1;     // the deep half of a pass is a chain: every phase reads the one before
    ++rec->nphases;
    return true;
  }
  // a point-wise GEMM launch: a phase when recording and plannable, else (after flushing what was recorded: order) its own launch
  void gemm(const GemmArgs& a) {
    if (rec) {
      WalkJob j;
      if (walk_plan_gemm(a, &j) && rec_push(j)) return;
      rec_flush();
    }
    launch_pw_gemm(a, st);
  }

  double* stats_slot(int groups) {
    double* p = c->stats + c->stats_top;
    c->stats_top += (size_t)clouds * groups * kGnWords;
    return p;
  }
  // the fp16 split of a weight matrix inside the context's blob (dsir_finalize_weights); off unless the split layers are on
  void split_of(GemmArgs& a) const {
    if (!c->agg_split || !c->dweights16 || a.W < c->dweights || a.W >= c->dweights + c->nweights) return;
    const size_t off = (size_t)(a.W - c->dweights);
    a.Wh = c->dweights16 + off;
    a.Wl = c->dweights16 + c->nweights + off;
  }
  static Seg seg_of(const Act& a, const int32_t* idx = nullptr, int64_t idx_cs = 0) {
    Seg s{};
    s.x = a.p; s.cloud_stride = (int64_t)a.rows * a.C; s.C = a.C; s.ld = a.C;
    s.idx = idx; s.idx_cloud_stride = idx_cs; s.gn = a.gn; s.act = a.act;
    s.uv = a.uv; s.uv_cloud_stride = (int64_t)(a.rows / kKnn) * 2 * a.C; s.dist = a.dist; s.dist_cloud_stride = a.rows; s.w8 = a.w8;
    return s;
  }
  // MLP2D: conv1x1 + GroupNorm (lazy) [+ LeakyReLU (lazy)]
  // out_buf / st_buf: caller-owned storage (persistent across launches) instead of the per-pass arenas
  Act mlp2d(const Mlp2dW& w, const Seg& s0, const Seg* s1, int M, bool act, float* out_buf = nullptr,
            double* st_buf = nullptr) {
    Act y;
    y.p = out_buf ? out_buf : c->ws.get<float>((size_t)clouds * M * w.cout);
    y.C = w.cout; y.rows = M; y.act = act ? 1 : 0;
    double* st_out = st_buf ? st_buf : stats_slot(w.groups);
    y.gn = GnRef{st_out, w.gamma, w.beta, w.groups, 1.0 / ((double)(w.cout / w.groups) * (double)M)};
    GemmArgs a;
    a.amode = A_SEGS; a.nseg = s1 ? 2 : 1; a.seg[0] = s0; if (s1) a.seg[1] = *s1;
    a.W = w.W; a.bias = w.b; a.Cin = w.cin; a.Cout = w.cout; a.M = M; a.clouds = clouds; a.epi = EPI_GN;
    a.Y = y.p; a.y_cloud_stride = (int64_t)M * w.cout; a.ldy = w.cout; a.stats_out = st_out; a.groups_out = w.groups;
    split_of(a);
    if (c->ws.overflow) return y;                 // an exhausted arena hands out its base: nothing may run on it
    if ((s0.uv && !s0.x) || (s1 && s1->uv && !s1->x)) {
      // table-only rows (lse_uv.hip) exist for ONE loader, pw_stream.hip's S_UV: the generic kernels would dereference the null row base
      rec_flush();
      if (!launch_pw_stream(a, st)) { c->sched_error = "MLP2D: no kernel took the table-only position encoding"; c->ws.overflow = true; }
      return y;
    }
    gemm(a);
    return y;
  }
  // mlp1 and mlp_skip of a block in ONE launch (same input; the weights one after the other, BlockW::pair_W): two outputs, two
  // sets of statistics - each element the chain the separate launch gives it.  False: not served (the caller launches them apart).
  bool mlp2d_pair(const BlockW& b, const Seg& s0, int M, Act& y1, Act& y2) {
    if (!b.pair_W) return false;
    const Mlp2dW &w1 = b.mlp1, &w2 = b.skip;
    GemmArgs a;
    a.amode = A_SEGS; a.nseg = 1; a.seg[0] = s0;
    a.W = b.pair_W; a.bias = b.pair_b; a.Cin = w1.cin; a.Cout = w1.cout + w2.cout; a.M = M; a.clouds = clouds; a.epi = EPI_GN;
    a.c_split = w1.cout;
    split_of(a);
    a.groups_out = w1.groups; a.groups_out2 = w2.groups;
    a.Y = reinterpret_cast<float*>(1); a.Y2 = a.Y; a.stats_out = reinterpret_cast<double*>(1); a.stats_out2 = a.stats_out;   // placeholders for the predicate
    a.ldy = w1.cout; a.ldy2 = w2.cout;
    if (!pw_gemm_serves_pair(a)) return false;
    y1.p = c->ws.get<float>((size_t)clouds * M * w1.cout); y1.C = w1.cout; y1.rows = M; y1.act = 1;
    y2.p = c->ws.get<float>((size_t)clouds * M * w2.cout); y2.C = w2.cout; y2.rows = M; y2.act = 0;
    double* st1 = stats_slot(w1.groups);
    double* st2 = stats_slot(w2.groups);
    y1.gn = GnRef{st1, w1.gamma, w1.beta, w1.groups, 1.0 / ((double)(w1.cout / w1.groups) * (double)M)};
    y2.gn = GnRef{st2, w2.gamma, w2.beta, w2.groups, 1.0 / ((double)(w2.cout / w2.groups) * (double)M)};
    a.Y = y1.p; a.y_cloud_stride = (int64_t)M * w1.cout; a.stats_out = st1;
    a.Y2 = y2.p; a.y2_cloud_stride = (int64_t)M * w2.cout; a.stats_out2 = st2;
    if (!c->ws.overflow) gemm(a);
    return true;
  }
  // lfa.mlp1 split by linearity (lse_uv.hip): per-point tables + dist + statistics, no output rows.  uv_buf / dist_buf: caller-owned
  // storage (persistent across launches) or nullptr
  Act lse_uv(const Mlp2dW& w, const float* w8, const float* xyz, int64_t xyz_cs, const int32_t* neigh, int64_t neigh_cs, int n,
             float* uv_buf, float* dist_buf, double* st_buf) {
    const int M = n * kKnn;
    Act y;
    y.p = nullptr; y.C = w.cout; y.rows = M; y.act = 1;
    float* uv = uv_buf ? uv_buf : c->ws.get<float>((size_t)clouds * n * 2 * w.cout);
    float* dist = dist_buf ? dist_buf : c->ws.get<float>((size_t)clouds * M);
    double* st_out = st_buf ? st_buf : stats_slot(w.groups);
    y.gn = GnRef{st_out, w.gamma, w.beta, w.groups, 1.0 / ((double)(w.cout / w.groups) * (double)M)};
    y.uv = uv; y.dist = dist; y.w8 = w8;
    LseUvArgs a;
    a.xyz = xyz; a.xyz_cs = xyz_cs; a.neigh = neigh; a.neigh_cs = neigh_cs; a.w8 = w8;
    a.uv = uv; a.uv_cs = (int64_t)n * 2 * w.cout; a.dist = dist; a.dist_cs = M;
    a.stats_out = st_out; a.groups = w.groups; a.n = n; a.clouds = clouds; a.KH = w.cout;
    rec_flush();
    if (!c->ws.overflow && !launch_lse_uv_stats(a, st)) { c->sched_error = "lse_uv: layer outside the kernel's envelope"; c->ws.overflow = true; }
    return y;
  }
  Act mlp2d_lse(const Mlp2dW& w, const float* xyz, int64_t xyz_cs, const int32_t* neigh, int64_t neigh_cs, int n,
                float* out_buf = nullptr, double* st_buf = nullptr) {
    const int M = n * kKnn;
    Act y;
    y.p = out_buf ? out_buf : c->ws.get<float>((size_t)clouds * M * w.cout);
    y.C = w.cout; y.rows = M; y.act = 1;
    double* st_out = st_buf ? st_buf : stats_slot(w.groups);
    y.gn = GnRef{st_out, w.gamma, w.beta, w.groups, 1.0 / ((double)(w.cout / w.groups) * (double)M)};
    GemmArgs a;
    a.amode = A_LSE; a.xyz = xyz; a.xyz_cloud_stride = xyz_cs; a.neigh = neigh; a.neigh_cloud_stride = neigh_cs;
    a.W = w.W; a.bias = w.b; a.Cin = 10; a.Cout = w.cout; a.M = M; a.clouds = clouds; a.epi = EPI_GN;
    a.Y = y.p; a.y_cloud_stride = (int64_t)M * w.cout; a.ldy = w.cout; a.stats_out = st_out; a.groups_out = w.groups;
    split_of(a);
    rec_flush();
    if (!c->ws.overflow) launch_pw_gemm(a, st);   // an exhausted arena hands out its base: nothing may run on it
    return y;
  }
  // Att_pooling up to (not including) its MLP2D: softmax_k(fc [gather(f); enc]) . [gather(f); enc]
  // s2 / s2_mode: optional cache of the enc half of the scores (kernels.h, GemmArgs::s2)
  Act att(const AttW& w, const Act& f, const Act& enc, const int32_t* neigh, int64_t neigh_cs, int n, float* s2 = nullptr,
          int s2_mode = 0) {
    Act y;
    y.p = c->ws.get<float>((size_t)clouds * n * w.d);
    y.C = w.d; y.rows = n;
    static const bool no_att2 = tuning_flag("DSIR_NO_ATT2");   // A/B switch
    if (att_pool_enabled() && (w.d == 64 || w.d == 128) && f.C * 2 == w.d && enc.C * 2 == w.d && (enc.p || w.d == 64) && c->dweights16 && w.fc >= c->dweights && w.fc < c->dweights + c->nweights &&
        !(s2 && s2_mode)) {
      // att_pool.hip, levels 1 / 2 unsplit: the whole score contraction on the matrix pipe, nothing gathered in the epilogue
      AttPool16Args a;
      a.f = f.p; a.f_cs = (int64_t)f.rows * f.C; a.f_ld = f.C; a.f_gn = f.gn; a.f_act = f.act;
      a.enc = enc.p; a.enc_cs = (int64_t)enc.rows * enc.C; a.enc_gn = enc.gn; a.enc_act = enc.act;
      a.uv = enc.uv; a.uv_cs = (int64_t)n * 2 * enc.C; a.dist = enc.dist; a.dist_cs = (int64_t)n * kKnn; a.w8 = enc.w8;
      a.neigh = neigh; a.neigh_cs = neigh_cs;
      const size_t off = (size_t)(w.fc - c->dweights);
      a.Wh = c->dweights16 + off; a.Wl = c->dweights16 + c->nweights + off; a.ldw = w.d;
      a.Y = y.p; a.y_cs = (int64_t)n * w.d; a.n = n; a.clouds = clouds;
      if (c->ws.overflow) return y;
      if (rec) {
        WalkJob j;
        if (walk_plan_att_full(a, w.d / 2, rec_wpc, &j) && rec_push(j)) return y;
        rec_flush();
      }
      if (launch_att_full(a, w.d / 2, st)) return y;
    }
    if (!enc.p && w.d >= 64) {     // table-only rows have no other consumer (lse_uv_enabled() excludes this)
      if (!c->ws.overflow) c->sched_error = "attentive pooling: no kernel took the table-only position encoding";
      c->ws.overflow = true;
      return y;
    }
    if (!no_att2 && w.d >= 64 && w.fc_g && f.C * 2 == w.d && enc.C * 2 == w.d) {   // d = 16: the extra gathers cost more than the MFMAs saved
      // score GEMM split by linearity: fc [gather(f); enc] = gather(W1 f) + W2 enc  (kernels.h, EPI_ATT2).
      // G = W1 f runs on n rows instead of 16 n; the pooling launch contracts only the enc half.
      float* G = c->ws.get<float>((size_t)clouds * n * w.d);
      GemmArgs g;
      g.amode = A_SEGS; g.nseg = 1; g.seg[0] = seg_of(f);
      g.W = w.fc_g; g.ldw = w.d / 2; g.bias = nullptr; g.Cin = w.d / 2; g.Cout = w.d; g.M = n; g.clouds = clouds;   // G in the consumer's column order (up_fc_g)
      g.epi = EPI_LINEAR; g.Y = G; g.y_cloud_stride = (int64_t)n * w.d; g.ldy = w.d;
      if (c->ws.overflow) return y;
      split_of(g);
      gemm(g);
      GemmArgs a2;
      a2.amode = A_SEGS; a2.nseg = 1; a2.seg[0] = seg_of(enc);
      a2.W = w.fc + w.d / 2; a2.ldw = w.d; a2.bias = nullptr; a2.Cin = w.d / 2; a2.Cout = w.d; a2.M = n * kKnn;
      a2.clouds = clouds; a2.epi = EPI_ATT2; a2.Y = y.p; a2.y_cloud_stride = (int64_t)n * w.d; a2.ldy = w.d;
      a2.g = G; a2.g_cloud_stride = (int64_t)n * w.d; a2.fseg = seg_of(f, neigh, neigh_cs);
      a2.s2 = s2; a2.s2_mode = s2 ? s2_mode : 0; a2.s2_cloud_stride = (int64_t)n * kKnn * w.d;
      split_of(a2);
      // G's column order is the consumer's (up_fc_g): d <= 128 belongs to pw_stream.hip, d = 256 to pw_tile.hip
      if (rec && w.d > 128) {
        WalkJob j;
        if (walk_plan_gemm(a2, &j) && rec_push(j)) return y;
      }
      rec_flush();
      if (w.d <= 128 ? launch_pw_stream(a2, st) : launch_pw_tile(a2, st)) return y;
    }
    if (att_pool_enabled() && w.d == 16 && f.C == 8 && enc.C == 8 && c->dweights16 && w.fc >= c->dweights && w.fc < c->dweights + c->nweights) {
      // att_pool.hip, level 0: four points per wave, fp16-split scores, softmax in registers
      AttPool16Args a;
      a.f = f.p; a.f_cs = (int64_t)f.rows * f.C; a.f_ld = f.C; a.f_gn = f.gn; a.f_act = f.act;
      a.enc = enc.p; a.enc_cs = (int64_t)enc.rows * enc.C; a.enc_gn = enc.gn; a.enc_act = enc.act;
      a.uv = enc.uv; a.uv_cs = (int64_t)n * 2 * enc.C; a.dist = enc.dist; a.dist_cs = (int64_t)n * kKnn; a.w8 = enc.w8;
      a.neigh = neigh; a.neigh_cs = neigh_cs;
      const size_t off = (size_t)(w.fc - c->dweights);
      a.Wh = c->dweights16 + off; a.Wl = c->dweights16 + c->nweights + off; a.ldw = w.d;
      a.Y = y.p; a.y_cs = (int64_t)n * w.d; a.n = n; a.clouds = clouds;
      rec_flush();
      if (!c->ws.overflow && launch_att_pool16(a, st)) return y;
    }
    if (!enc.p) {     // table-only rows have no other consumer (lse_uv_enabled() excludes this)
      if (!c->ws.overflow) c->sched_error = "attentive pooling: no kernel took the table-only position encoding";
      c->ws.overflow = true;
      return y;
    }
    GemmArgs a;
    a.amode = A_SEGS; a.nseg = 2;
    a.seg[0] = seg_of(f, neigh, neigh_cs);
    a.seg[1] = seg_of(enc);
    a.W = w.fc; a.bias = nullptr; a.Cin = w.d; a.Cout = w.d; a.M = n * kKnn; a.clouds = clouds; a.epi = EPI_ATT;
    a.Y = y.p; a.y_cloud_stride = (int64_t)n * w.d; a.ldy = w.d;
    split_of(a);
    rec_flush();
    if (!c->ws.overflow) launch_pw_gemm(a, st);   // an exhausted arena hands out its base: nothing may run on it
    return y;
  }
  Act linear(const LinW& w, const Seg& s0, const Seg* s1, int M, int epi, float* out = nullptr,
             const float* residual = nullptr) {
    Act y;
    y.p = out ? out : c->ws.get<float>((size_t)clouds * M * w.cout);
    y.C = w.cout; y.rows = M;
    GemmArgs a;
    a.amode = A_SEGS; a.nseg = s1 ? 2 : 1; a.seg[0] = s0; if (s1) a.seg[1] = *s1;
    a.W = w.W; a.bias = w.b; a.Cin = w.cin; a.Cout = w.cout; a.M = M; a.clouds = clouds; a.epi = epi;
    a.Y = y.p; a.y_cloud_stride = (int64_t)M * w.cout; a.ldy = w.cout;
    a.residual = residual; a.res_cloud_stride = (int64_t)M * w.cout; a.ldres = w.cout;
    split_of(a);
    rec_flush();
    if (!c->ws.overflow) launch_pw_gemm(a, st);   // an exhausted arena hands out its base: nothing may run on it
    return y;
  }
};

Seg plain_seg(const float* x, int64_t cloud_stride, int C, int ld, const int32_t* idx = nullptr, int64_t idx_cs = 0) {
  Seg s{};
  s.x = x; s.cloud_stride = cloud_stride; s.C = C; s.ld = ld; s.idx = idx; s.idx_cloud_stride = idx_cs;
  s.gn = GnRef{nullptr, nullptr, nullptr, 0, 0.0}; s.act = 0;
  return s;
}

// RandLA.forward (RandLANet.py:311-372).  in0/in1: the (possibly concatenated / gathered) input features.
// The position-encoding branch of every level (lfa.mlp1 on the relative position code, lfa.mlp2 on top of it)
// depends only on the pyramid and the weights.  The inlier model runs on the SAME (src) pyramid in every
// registration iteration (model.py:575), so that branch is computed in iteration 0 into caller-owned
// storage and re-used afterwards: same kernels, same inputs, same bits (SURVEY §7.2 loop invariants).
struct EncCache {
  bool valid = false;
  float* s2_buf[DSIR_MAX_LEVELS][2] = {};   // enc half of the attention scores (W2 enc / W2 enc2) of the split levels (d >= 64)
  float* enc_buf[DSIR_MAX_LEVELS] = {};
  float* uv_buf[DSIR_MAX_LEVELS] = {};     // levels whose lfa.mlp1 rows are not stored (lse_uv.hip): tables + dist instead of enc_buf
  float* dist_buf[DSIR_MAX_LEVELS] = {};
  float* enc2_buf[DSIR_MAX_LEVELS] = {};
  double* enc_stats[DSIR_MAX_LEVELS] = {};
  double* enc2_stats[DSIR_MAX_LEVELS] = {};
  Act enc[DSIR_MAX_LEVELS], enc2[DSIR_MAX_LEVELS];
};

int randla_forward(dsir_ctx* c, const RandlaW& w, const Seg& in0, const Seg* in1, const Pyramid& py, float* feat_out,
                   float* logits_out, EncCache* cache = nullptr) {
  const dsir_cfg& g = c->cfg;
  const int L = g.num_layers;
  hipStream_t st = c->stream;
  Sched s{c, st, py.clouds};
  // 34 GroupNorm layers x clouds x <=8 groups x kGnWords words
  const size_t stats_need = (size_t)40 * py.clouds * 8 * kGnWords;
  if (c->stats_prezeroed && c->stats_base + stats_need <= c->stats_cap) {
    c->stats_top = c->stats_base;               // zeroed by register_enqueue together with the other passes' regions
    c->stats_base += stats_need;
  } else {
    c->stats_top = 0;
    if (stats_need > c->stats_cap) return fail(c, "stats arena too small (%zu > %zu)", stats_need, c->stats_cap);
    HIP_OK(c, hipMemsetAsync(c->stats, 0, stats_need * sizeof(double), st));
  }

  const int64_t xyz_cs = (int64_t)py.S * 3, neigh_cs = (int64_t)py.S * kKnn, sub_cs = (int64_t)py.S1 * kKnn, interp_cs = py.S;
  // Deep-level walker (walk.hip): with a few clouds in flight the layers from level 1's pooling down to the decoder block of level 2
  // run as phases of ONE launch.  Which launches become phases is decided layer by layer (Sched::gemm / att: the same kernels, the
  // same bits); the position-encoding branch of the deep levels (lfa.mlp1, lfa.mlp2: row-streaming kernels the walker does not hold,
  // inputs the pyramid alone) is computed ahead of the chain so that it does not cut the chain in pieces.
  static const int walk_from = 2;                            // first level inside the walker
  const bool walk = c->walk_mode && c->walk_dev && py.clouds <= dsir_ctx::kWalkClouds && L > walk_from &&
                    c->walk_used < dsir_ctx::kWalkSlots;
  WalkProgram wprog;
  wprog.clouds = py.clouds;
  wprog.flags = c->walk_flags;
  wprog.wpc = c->walk_wpc > 0 ? c->walk_wpc : (py.clouds <= 8 ? 32 : 16);   // 256 workgroups: one per CU (the walker holds the widest bodies' registers)
  const bool reuse = cache && cache->valid;
  auto enc_of = [&](int l) {      // lfa.mlp1 of level l (RandLANet.py:176-177): per-point tables (levels 0 / 1) or the stored rows
    const BlockW& b = w.blk[l];
    const int n = py.nl[l];
    const float* xyz_l = py.xyz + (int64_t)py.off[l] * 3;
    const int32_t* nb_l = py.neigh + (int64_t)py.off[l] * kKnn;
    const bool uvl = b.lse_w8 && lse_uv_enabled();     // this level's lfa.mlp1 rows are rebuilt from per-point tables, never stored
    return reuse ? cache->enc[l]
           : uvl ? s.lse_uv(b.lfa1, b.lse_w8, xyz_l, xyz_cs, nb_l, neigh_cs, n, cache ? cache->uv_buf[l] : nullptr,
                            cache ? cache->dist_buf[l] : nullptr, cache ? cache->enc_stats[l] : nullptr)
                 : s.mlp2d_lse(b.lfa1, xyz_l, xyz_cs, nb_l, neigh_cs, n, cache ? cache->enc_buf[l] : nullptr,
                               cache ? cache->enc_stats[l] : nullptr);
  };
  auto enc2_of = [&](int l, const Act& enc) {   // lfa.mlp2 on top of it (RandLANet.py:186)
    const BlockW& b = w.blk[l];
    const int n = py.nl[l];
    const int32_t* nb_l = py.neigh + (int64_t)py.off[l] * kKnn;
    return reuse ? cache->enc2[l]
                 : s.mlp2d(b.lfa2, Sched::seg_of(enc, enc.uv ? nb_l : nullptr, enc.uv ? neigh_cs : 0), nullptr, n * kKnn, true,
                           cache ? cache->enc2_buf[l] : nullptr, cache ? cache->enc2_stats[l] : nullptr);
  };
  Act enc_pre[DSIR_MAX_LEVELS], enc2_pre[DSIR_MAX_LEVELS];
  Act x = s.mlp2d(w.pre, in0, in1, py.nl[0], true);
  std::vector<Act> skips;
  for (int l = 0; l < L; ++l) {
    const BlockW& b = w.blk[l];
    const int n = py.nl[l];
    const int32_t* nb_l = py.neigh + (int64_t)py.off[l] * kKnn;
    const Seg xin = Sched::seg_of(x);
    Act f, skipb;
    const bool ahead = walk && l >= walk_from;       // computed before the chain started (below)
    // Few clouds in flight: the block's independent branches leave the chain mlp1 -> pooling 1 -> pooling 2 -> mlp2 and run beside it
    // on the auxiliary streams - the position-encoding branch (lfa.mlp1 -> lfa.mlp2: inputs the pyramid alone), joined where the two
    // poolings read it, and mlp_skip (input the block's input), joined at the residual sum.
    const bool fork_enc = fork_on(c, py.clouds, 2) && !walk && !reuse && !ahead, fork_skip = fork_on(c, py.clouds, 4) && !walk;
    hipEvent_t ev_enc = nullptr, ev_enc2 = nullptr, ev_skip = nullptr;
    const hipEvent_t block_in = (fork_enc || fork_skip) ? fork_mark(c, st) : nullptr;      // x (and everything before it) is in flight on the main stream
    Act enc, enc2;
    if (fork_enc) {
      fork_wait(c->aux[0], block_in);
      s.st = c->aux[0];
      enc = enc_of(l); ev_enc = fork_mark(c, c->aux[0]);
      enc2 = enc2_of(l, enc); ev_enc2 = fork_mark(c, c->aux[0]);
      s.st = st;
    }
    const bool paired = s.mlp2d_pair(b, xin, n, f, skipb);
    if (!paired) {
      f = s.mlp2d(b.mlp1, xin, nullptr, n, true);
      if (fork_skip) {
        fork_wait(c->aux[1], block_in);
        s.st = c->aux[1];
        skipb = s.mlp2d(b.skip, xin, nullptr, n, false);
        ev_skip = fork_mark(c, c->aux[1]);
        s.st = st;
      }
    }
    if (!ev_enc) enc = ahead ? enc_pre[l] : enc_of(l);
    else fork_wait(st, ev_enc);
    const int s2_mode = reuse ? 2 : 1;      // iteration 0 stores the pyramid-only half of the scores, later iterations load it
    Act agg = s.att(b.att1, f, enc, nb_l, neigh_cs, n, cache ? cache->s2_buf[l][0] : nullptr, s2_mode);
    Act a1 = s.mlp2d(b.att1.mlp, Sched::seg_of(agg), nullptr, n, true);
    if (!ev_enc2) enc2 = ahead ? enc2_pre[l] : enc2_of(l, enc);
    else fork_wait(st, ev_enc2);
    if (cache && !reuse) { cache->enc[l] = enc; cache->enc2[l] = enc2; }
    Act agg2 = s.att(b.att2, a1, enc2, nb_l, neigh_cs, n, cache ? cache->s2_buf[l][1] : nullptr, s2_mode);
    Act a2 = s.mlp2d(b.att2.mlp, Sched::seg_of(agg2), nullptr, n, true);
    Act mainb = s.mlp2d(b.mlp2, Sched::seg_of(a2), nullptr, n, false);
    if (!paired) {
      if (ev_skip) fork_wait(st, ev_skip);
      else skipb = s.mlp2d(b.skip, xin, nullptr, n, false);
    }
    Act enc_out;
    enc_out.C = 2 * b.d; enc_out.rows = n;
    Act samp;
    samp.C = enc_out.C; samp.rows = py.nl[l + 1];
    samp.p = c->ws.get<float>((size_t)py.clouds * samp.rows * samp.C);
    if (l == 0) enc_out.p = c->ws.get<float>((size_t)py.clouds * n * enc_out.C);
    if (c->ws.overflow) {
      if (c->sched_error) { const char* m = c->sched_error; c->sched_error = nullptr; return fail(c, "randla_forward: %s (level %d)", m, l); }
      return fail(c, "workspace exhausted in randla_forward (raise max_points / max_pairs)");
    }
    if (l == 0) {
      // the level-0 block output is also the decoder's last skip connection: materialise it
      launch_residual_combine(mainb.p, mainb.gn, skipb.p, skipb.gn, enc_out.C, n, py.clouds, enc_out.p, st);
      launch_gather_max(enc_out.p, (int64_t)n * enc_out.C, py.sub + (int64_t)py.soff[l] * kKnn, sub_cs, samp.C, samp.rows,
                        py.clouds, samp.p, st);
    } else {
      if (walk && l == walk_from - 1) {
        // the chain starts with this level's pooling: first the deep levels' position-encoding branch, as launches of their own
        for (int q = walk_from; q < L; ++q) { enc_pre[q] = enc_of(q); enc2_pre[q] = enc2_of(q, enc_pre[q]); }
        if (c->ws.overflow) return fail(c, "workspace exhausted in randla_forward (raise max_points / max_pairs)");
        s.rec = &wprog; s.rec_wpc = wprog.wpc;
      }
      // deeper levels: only the pooled ("randomly sampled") rows are ever read — combine inside the pooling kernel
      GmcArgs ga{mainb.p, mainb.gn, skipb.p, skipb.gn, n, py.sub + (int64_t)py.soff[l] * kKnn, sub_cs, samp.C, samp.rows, samp.p, 0};
      WalkJob wj;
      if (!(s.rec && walk_plan_gmc(ga, s.rec_wpc, &wj) && s.rec_push(wj))) {
        s.rec_flush();
        launch_gather_max_combine(mainb.p, mainb.gn, skipb.p, skipb.gn, n, py.sub + (int64_t)py.soff[l] * kKnn, sub_cs,
                                  samp.C, samp.rows, py.clouds, samp.p, st);
      }
    }
    if (l == 0) skips.push_back(enc_out);
    skips.push_back(samp);
    x = samp;
  }
  x = s.mlp2d(w.mid, Sched::seg_of(skips.back()), nullptr, py.nl[L], true);
  for (int j = 0; j < L; ++j) {
    const int lvl = L - 1 - j;
    const Act& sk = skips[skips.size() - 2 - j];
    const Seg s0 = Sched::seg_of(sk);
    const Seg s1 = Sched::seg_of(x, py.interp + py.off[lvl], interp_cs);
    if (s.rec && lvl < walk_from) { s.rec_flush(); s.rec = nullptr; }      // the chain ends with the decoder block of level walk_from
    x = s.mlp2d(w.dec[j], s0, &s1, py.nl[lvl], true);
  }
  if (s.rec) { s.rec_flush(); s.rec = nullptr; }
  if (s.rec_error) return 1;
  const int n0 = py.nl[0];
  bool fused = false;
  static const bool no_head = tuning_flag("DSIR_NO_HEAD");   // A/B switch
  if (logits_out && !no_head &&w.dec_out == 32 && g.out_feat_dim == 64 && w.fc[0].cout == 64 && w.fc[1].cout == 32) {
    // mlp_out + fc_label in one launch (head_mlp.hip); bit-identical to the four launches below
    HeadArgs h;
    h.in = Sched::seg_of(x);
    h.W1 = w.out_w; h.W2 = w.fc[0].W; h.b2 = w.fc[0].b; h.W3 = w.fc[1].W; h.b3 = w.fc[1].b; h.W4 = w.fc[2].W; h.b4 = w.fc[2].b;
    h.ncls = w.ncls; h.M = n0; h.clouds = py.clouds; h.feat_out = feat_out; h.logits_out = logits_out;
    if (c->ws.overflow) return fail(c, "workspace exhausted in randla_forward (raise max_points / max_pairs)");
    // default: the head's four layers as fp16-split products (head_mlp_h.hip; fp32 accuracy); dsir_enable_agg_split(0) /
    // DSIR_AGG_F32: the exact-fp32 head, bit-identical to the four separate launches
    if (c->agg_split) {
      for (int k = 0; k < 4; ++k) { h.Wh[k] = w.head_wh[k]; h.Wl[k] = w.head_wl[k]; }
      fused = launch_head_mlp_h(h, st);
    }
    if (!fused) fused = launch_head_mlp(h, st);
  }
  LinW ow; ow.W = w.out_w; ow.b = nullptr; ow.cin = w.dec_out; ow.cout = g.out_feat_dim;
  Act feat;
  if (!fused) feat = s.linear(ow, Sched::seg_of(x), nullptr, n0, EPI_LINEAR, feat_out);
  if (logits_out && !fused) {
    Act h = s.linear(w.fc[0], Sched::seg_of(feat), nullptr, n0, EPI_ACT);
    h = s.linear(w.fc[1], Sched::seg_of(h), nullptr, n0, EPI_ACT);
    s.linear(w.fc[2], Sched::seg_of(h), nullptr, n0, EPI_LINEAR, logits_out);
  }
  if (c->ws.overflow) return fail(c, "workspace exhausted in randla_forward (raise max_points / max_pairs)");
  if (cache) cache->valid = true;
  return 0;
}

// mlp_feat (loop invariant part of Network.aggregation, model.py:218)
float* run_mlp_feat(dsir_ctx* c, const float* feat0, int clouds, int n, float* out = nullptr) {   // out: caller-owned [clouds][n][64] or the arena
  Sched s{c, c->stream, clouds};
  const NetW& w = c->net;
  Act h = s.linear(w.mlp_feat[0], plain_seg(feat0, (int64_t)n * 64, 64, 64), nullptr, n, EPI_ACT);
  h = s.linear(w.mlp_feat[1], Sched::seg_of(h), nullptr, n, EPI_ACT);
  h = s.linear(w.mlp_feat[2], Sched::seg_of(h), nullptr, n, EPI_LINEAR, out);
  return h.p;
}
// normalize(mlp_proj(F + mlp_att([xyz; score])))   (model.py:223-234)
// what the descriptor search needs of the descriptors besides their values (AggArgs: sq, hi / lo, packed_init); the fp16-split chain
// writes them in its epilogue and returns true, any other path leaves them to the search's own preparation kernels
struct AggExtras { float* sq = nullptr; void* hi = nullptr; void* lo = nullptr; unsigned long long* packed_init = nullptr; };
bool run_att_proj(dsir_ctx* c, const float* xyz, int64_t xyz_cs, const float* score, const float* F, int clouds, int n,
                  float* desc, const AggExtras* ex = nullptr) {
  Sched s{c, c->stream, clouds};
  const NetW& w = c->net;
  static const bool no_agg = tuning_flag("DSIR_NO_AGG");   // A/B switch
  const LinW* m = w.mlp_att;
  if (!no_agg && m[0].cin == 4 && m[0].cout == 32 && m[1].cout == 64 && m[2].cout == 128 && m[3].cout == 256 &&
      m[4].cout == 64 && w.mlp_proj.cin == 64 && w.mlp_proj.cout == 64) {
    AggArgs a;
    a.xyz = xyz; a.xyz_cs = xyz_cs; a.score = score; a.F = F;
    a.W1 = m[0].W; a.b1 = m[0].b; a.W2 = m[1].W; a.b2 = m[1].b; a.W3 = m[2].W; a.b3 = m[2].b;
    a.W4 = m[3].W; a.b4 = m[3].b; a.W5 = m[4].W; a.b5 = m[4].b; a.W6 = w.mlp_proj.W; a.b6 = w.mlp_proj.b;
    a.desc = desc; a.n = n; a.clouds = clouds;
    // default: the chain's wide layers as fp16-split products on the fp16 matrix pipe (agg_chain_h.hip: fp32 accuracy, not the
    // fp32 kernel's bits); dsir_enable_agg_split(0) / DSIR_AGG_F32: the exact-fp32 chain, bit-identical to the unfused launches below
    static const bool no_fuse = tuning_flag("DSIR_NO_AGG_EXTRAS");   // A/B switch: the search prepares its operands itself
    if (c->agg_split) {
      for (int k = 0; k < 5; ++k) { a.Wh[k] = c->agg_wh[k]; a.Wl[k] = c->agg_wl[k]; }
      if (ex && !no_fuse) { a.sq = ex->sq; a.hi = ex->hi; a.lo = ex->lo; a.packed_init = ex->packed_init; }
      if (launch_agg_chain_h(a, c->stream)) return ex && !no_fuse;
      a.sq = nullptr; a.hi = a.lo = nullptr; a.packed_init = nullptr;
    }
    if (launch_agg_chain(a, c->stream)) return false;
  }
  const Seg sx = plain_seg(xyz, xyz_cs, 3, 3);
  const Seg ss = plain_seg(score, n, 1, 1);
  Act h = s.linear(w.mlp_att[0], sx, &ss, n, EPI_ACT);
  for (int k = 1; k < 4; ++k) h = s.linear(w.mlp_att[k], Sched::seg_of(h), nullptr, n, EPI_ACT);
  h = s.linear(w.mlp_att[4], Sched::seg_of(h), nullptr, n, EPI_LINEAR, nullptr, F);
  s.linear(w.mlp_proj, Sched::seg_of(h), nullptr, n, EPI_L2NORM, desc);
  return false;
}

int build_pyramid(dsir_ctx* c, const float* points, int stride, int clouds, int n, float* xyz, int32_t* neigh,
                  int32_t* sub, int32_t* interp) {
  const dsir_cfg& g = c->cfg;
  Pyramid p;
  fill_pyramid_layout(g, clouds, n, p);
  if (p.nl[g.num_layers - 1] < kKnn)
    return fail(c, "cloud too small: level %d has %d < %d points (need n >= %d)", g.num_layers - 1,
                p.nl[g.num_layers - 1], kKnn, kKnn * 64);
  hipStream_t st = c->stream;
  const int64_t xyz_cs = (int64_t)p.S * 3, neigh_cs = (int64_t)p.S * kKnn, sub_cs = (int64_t)p.S1 * kKnn;
  PyramidLevels lv{};
  lv.L = g.num_layers; lv.S = p.S; lv.S1 = p.S1;
  for (int l = 0; l <= g.num_layers; ++l) { lv.nl[l] = p.nl[l]; lv.off[l] = p.off[l]; lv.soff[l] = p.soff[l]; }
  // every level's points are a prefix of the level above, hence of the input cloud (data_base.py:166-172): one launch for all
  launch_copy_xyz_levels(points, (int64_t)n * stride, stride, lv, clouds, xyz, xyz_cs, st);
  static const bool no_grid = tuning_flag("DSIR_NO_GRID");   // A/B switch
  static const int grid_min = (int)tuning_int("DSIR_GRID_MIN", 1024);   // tuning hook
  static const bool no_nn1_grid = tuning_flag("DSIR_NO_NN1_GRID");   // A/B switch: brute-force interpolation search throughout
  static const long long nn1_grid_min = tuning_int("DSIR_NN1_GRID_MIN", 65536);   // tuning hook
  // interpolation search of level l (support = level l + 1) through level l + 1's grid: when that level has one and the launch has
  // queries enough to fill the chip with one lane per query (same bits either way)
  auto nn1_by_grid = [&](int l) {
    return !no_grid && !no_nn1_grid && l + 1 < g.num_layers && p.nl[l + 1] >= grid_min && (int64_t)clouds * p.nl[l] >= nn1_grid_min;
  };
  // the grid scratch of every level stays until the pyramid is done: the level above's sorted points are the QUERIES of this level's
  // interpolation search (in cell order: a wave's lanes walk neighbouring cells)
  const size_t mark = c->ws.mark();
  const void* prev_scratch = nullptr;
  bool any_nn1_grid = false;
  for (int l = 0; l < g.num_layers; ++l) any_nn1_grid = any_nn1_grid || nn1_by_grid(l);
  if (fork_on(c, clouds, 1) && !any_nn1_grid && g.num_layers == 4) {
    // few clouds: the searches of the four levels (and the four interpolation searches) read the input points alone - three branches
    // of about equal length instead of a chain of eight launches: {level 0} | {level 1, its interpolation search} | {the rest}
    hipStream_t a0 = c->aux[0], a1 = c->aux[1];
    const hipEvent_t start = fork_mark(c, st);
    fork_wait(a0, start); fork_wait(a1, start);
    auto knn_level = [&](int l, hipStream_t s_) -> bool {
      if (p.nl[l] >= grid_min && !no_grid) {
        void* scratch = c->ws.raw(knn_grid_scratch_bytes(clouds, p.nl[l]));
        if (c->ws.overflow) return false;
        launch_knn16_grid(points, (int64_t)n * stride, stride, p.nl[l], clouds, neigh + (int64_t)p.off[l] * kKnn, neigh_cs, scratch, s_);
      } else {
        launch_knn16(points, (int64_t)n * stride, stride, p.nl[l], clouds, neigh + (int64_t)p.off[l] * kKnn, neigh_cs, s_);
      }
      return true;
    };
    auto nn1_level = [&](int l, hipStream_t s_) {
      launch_nn1(points, (int64_t)n * stride, stride, p.nl[l], p.nl[l + 1], clouds, interp + p.off[l], p.S, s_);
    };
    bool ok = knn_level(0, st);
    ok = ok && knn_level(1, a0);
    nn1_level(1, a0);
    nn1_level(0, a1);
    ok = ok && knn_level(2, a1) && knn_level(3, a1);
    nn1_level(2, a1); nn1_level(3, a1);
    fork_to(c, a0, st); fork_to(c, a1, st);          // join before anything re-uses the scratch or reads the lists
    if (!ok) return fail(c, "workspace exhausted in the KNN pyramid");
    c->ws.release(mark);
    launch_copy_sub_levels(neigh, neigh_cs, lv, clouds, sub, sub_cs, st);
    return 0;
  }
  // Every level's searches read the input points alone (the levels are prefixes of the cloud), so the pyramid is THREE launches instead
  // of a chain of ten - the grids of the large levels, their searches, and everything else (the interpolation searches of all levels,
  // the 16-NN of the levels without a grid) - plus one per interpolation search that walks a grid (large launches).  With one pair in
  // flight (the reference's evaluation mode, test.py:56) the chain was 226 us of the registration's 3.05 ms, now 135; with eight, 373.
  // The same kernels' bodies on the same operands: same bits (tests/test_gpu_parity.py, two-process A/B).
  static const bool no_merge = tuning_flag("DSIR_NO_PYRAMID_MERGE");   // A/B switch
  if (!no_merge && g.num_layers <= KnnSmallJobs::kMax / 2) {
    int ngrid = 0, gn[4], grid_of[8];
    int32_t* gout[4];
    KnnSmallJobs jobs{};
    bool ok = true;
    for (int l = 0; l < g.num_layers && ok; ++l) {
      grid_of[l] = -1;
      if (p.nl[l] >= grid_min && !no_grid) {
        ok = knn16_grid_can_merge(p.nl[l]) && ngrid < 4;
        if (ok) { gn[ngrid] = p.nl[l]; gout[ngrid] = neigh + (int64_t)p.off[l] * kKnn; grid_of[l] = ngrid++; }
      } else {
        jobs.job[jobs.njobs++] = {knn16_takes_wave_kernel(p.nl[l], clouds) ? 1 : 2, p.nl[l], 0, neigh + (int64_t)p.off[l] * kKnn, neigh_cs, 0};
      }
      if (!nn1_by_grid(l)) jobs.job[jobs.njobs++] = {0, p.nl[l], p.nl[l + 1], interp + p.off[l], (int64_t)p.S, 0};
    }
    if (ok) {
      void* gscr[4];
      for (int k = 0; k < ngrid; ++k) gscr[k] = c->ws.raw(knn_grid_scratch_bytes(clouds, gn[k]));
      if (c->ws.overflow) return fail(c, "workspace exhausted in the KNN pyramid");
      if (ngrid) launch_knn16_grid_levels(points, (int64_t)n * stride, stride, ngrid, gn, clouds, gout, neigh_cs, gscr, st);
      launch_knn_small_levels(points, (int64_t)n * stride, stride, clouds, jobs, st);
      for (int l = 0; l + 1 < g.num_layers; ++l)
        if (nn1_by_grid(l))      // level l + 1 has a grid (nn1_by_grid): the search walks it, its queries in level l's cell order when that has one
          launch_nn1_grid(points, (int64_t)n * stride, stride, p.nl[l], p.nl[l + 1], clouds, interp + p.off[l], p.S, gscr[grid_of[l + 1]], st,
                          grid_of[l] >= 0 ? gscr[grid_of[l]] : nullptr);
      c->ws.release(mark);
      launch_copy_sub_levels(neigh, neigh_cs, lv, clouds, sub, sub_cs, st);
      return 0;
    }
  }
  for (int l = 0; l < g.num_layers; ++l) {
    if (p.nl[l] >= grid_min && !no_grid) {
      // large levels: exact grid-pruned search (knn_grid.hip); same bits as the brute force
      void* scratch = c->ws.raw(knn_grid_scratch_bytes(clouds, p.nl[l]));
      if (c->ws.overflow) return fail(c, "workspace exhausted in the KNN pyramid");
      launch_knn16_grid(points, (int64_t)n * stride, stride, p.nl[l], clouds, neigh + (int64_t)p.off[l] * kKnn, neigh_cs,
                        scratch, st);
      // this level's points are the support of the level above's interpolation search: it walks the grid just built
      if (l > 0 && nn1_by_grid(l - 1))
        launch_nn1_grid(points, (int64_t)n * stride, stride, p.nl[l - 1], p.nl[l], clouds, interp + p.off[l - 1], p.S, scratch, st,
                        prev_scratch);
      prev_scratch = scratch;
    } else {
      prev_scratch = nullptr;
      launch_knn16(points, (int64_t)n * stride, stride, p.nl[l], clouds, neigh + (int64_t)p.off[l] * kKnn, neigh_cs, st);
    }
    if (!nn1_by_grid(l)) launch_nn1(points, (int64_t)n * stride, stride, p.nl[l], p.nl[l + 1], clouds, interp + p.off[l], p.S, st);
  }
  c->ws.release(mark);   // stream-ordered: later users of this memory run after the query kernels
  // sub_idx of level l = the neighbour lists of its first n_{l+1} points: all levels in one launch
  launch_copy_sub_levels(neigh, neigh_cs, lv, clouds, sub, sub_cs, st);
  return 0;
}

int check_ready(dsir_ctx* c) {
  if (!c) return 1;
  if (!c->finalized) return fail(c, "weights not finalized (call dsir_load_weight for every key, then dsir_finalize_weights)");
  c->ws.top = 0; c->ws.overflow = false; c->sched_error = nullptr;
  return 0;
}

int post(dsir_ctx* c) {
  if (c->ws.overflow) return fail(c, "workspace exhausted (raise max_points / max_pairs in dsir_cfg)");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, "HIP launch error: %s", hipGetErrorString(e));
  return 0;
}

dsir_ctx::MatchEvents* match_event_slot(dsir_ctx* c) {
  if (!c->time_match) return nullptr;
  if (c->match_events_used == c->match_events.size()) {
    dsir_ctx::MatchEvents e{};
    hipEventCreate(&e.op0); hipEventCreate(&e.op1); hipEventCreate(&e.k0); hipEventCreate(&e.k1);
    c->match_events.push_back(e);
  }
  return &c->match_events[c->match_events_used++];
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int dsir_create(int device, const dsir_cfg* cfg, dsir_ctx** out) {
  if (!cfg || !out) return fail(nullptr, "dsir_create: null argument");
  if (cfg->num_knn != kKnn) return fail(nullptr, "num_knn must be %d (got %d)", kKnn, cfg->num_knn);
  if (cfg->num_layers != 4) return fail(nullptr, "num_layers must be 4 (got %d)", cfg->num_layers);
  if (cfg->out_feat_dim != 64) return fail(nullptr, "out_feat_dim must be 64 (got %d)", cfg->out_feat_dim);
  if (cfg->num_classes < 1 || cfg->num_classes > 32) return fail(nullptr, "num_classes out of range");
  for (int l = 0; l < 4; ++l) {
    if (cfg->d_out[l] % 16 || cfg->d_out[l] < 16 || cfg->d_out[l] > 256) return fail(nullptr, "d_out[%d]=%d unsupported", l, cfg->d_out[l]);
    if (cfg->sub_sampling_ratio[l] < 1) return fail(nullptr, "bad sub_sampling_ratio");
  }
  if (cfg->feat_len < 3 || cfg->feat_len > 16) return fail(nullptr, "feat_len must be in [3,16]");
  if (cfg->max_points < kKnn * 64 || cfg->max_pairs < 1) return fail(nullptr, "max_points must be >= %d and max_pairs >= 1", kKnn * 64);
  // several kernels address a cloud's rows with 32-bit byte offsets from a per-cloud base (n * 16 * d * 4 < 2^32 at d = 64: knn_grid.hip,
  // att_pool.hip, lse_uv.hip check their own products); 2^20 points per cloud is far beyond what the 2.5 kB-per-point workspace admits
  if (cfg->max_points > (1 << 20)) return fail(nullptr, "max_points must be <= %d", 1 << 20);
  // the GroupNorm statistics' atomics are exact - order independent - for at most kGnMaxContrib contributions per statistic
  // (device_utils.h); the largest layer of a cloud of max_points points must stay within that
  if (const int gc = gn_max_contributions(*cfg, cfg->max_points); gc > kGnMaxContrib)
    return fail(nullptr, "max_points=%d: %d workgroup contributions per GroupNorm statistic exceed the exactness bound %d of the statistics' atomics (max_points <= %d)",
                cfg->max_points, gc, kGnMaxContrib, dsir_max_points_limit(cfg));
  if (cfg->pipeline < DSIR_PIPELINE_ALIGN || cfg->pipeline > DSIR_PIPELINE_LABEL) return fail(nullptr, "unknown pipeline %d", cfg->pipeline);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(nullptr, "device %d out of range (%d devices)", device, ndev);
  dsir_ctx* c = new dsir_ctx();
  c->device = device; c->cfg = *cfg;
  c->screen_mode = tuning_flag("DSIR_NO_SCREEN") ? 0 : 1;
  if (const char* e = tuning_env("DSIR_PRUNE_MIN_K")) c->prune_min_points = atoi(e) > 0 ? atoi(e) : 0;   // A/B hook; 0 = off
  if (const char* e = tuning_env("DSIR_PRUNE_MIN_ROWS")) c->prune_min_rows = atoll(e) > 0 ? atoll(e) : 0;   // tuning hook
  c->agg_split = tuning_flag("DSIR_AGG_F32") ? 0 : 1;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(nullptr, "cannot initialise device %d", device);
  }
  c->own_stream = c->stream;
  c->fork_mode = tuning_flag("DSIR_FORK") ? 1 : 0;
  if (c->fork_mode) ensure_aux_streams(c);
  // which sub-networks exist follows args.pipeline (model.py:131-193)
  add_randla(c, "feat_extractor", cfg->feat_len, cfg->num_classes);
  if (cfg->pipeline != DSIR_PIPELINE_LABEL) {
    add_mlp1d(c, "mlp_feat", {64, 64, 128, 64});
    add_mlp1d(c, "mlp_att", {4, 32, 64, 128, 256, 64});
    add_mlp1d(c, "mlp_proj", {64, 64});
  }
  if (cfg->pipeline == DSIR_PIPELINE_ALIGN) add_randla(c, "inlier_model", 6, 1);
  // workspace: ~1.4k floats per point per cloud for one RandLA pass, 2P clouds, plus per-pair state
  const size_t clouds = (size_t)2 * cfg->max_pairs;
  const size_t per_cloud = (size_t)cfg->max_points * 2560 * sizeof(float) + ((size_t)1 << 22);
  c->ws.cap = clouds * per_cloud + ((size_t)64 << 20);
  if (cfg->pipeline == DSIR_PIPELINE_ALIGN) {
    // per pair: the inlier model's cached enc halves of the attention scores (EncCache::s2_buf), 2 x n_l x 16 x d_l floats
    size_t s2 = 0;
    int nl = cfg->max_points;
    for (int l = 0; l < cfg->num_layers; ++l) {
      if (cfg->d_out[l] >= 64) s2 += (size_t)2 * nl * kKnn * cfg->d_out[l] * sizeof(float) + 512;
      nl /= cfg->sub_sampling_ratio[l];
    }
    c->ws.cap += (size_t)cfg->max_pairs * s2;
  }
  if (hipMalloc((void**)&c->ws.base, c->ws.cap) != hipSuccess) {
    const size_t cap = c->ws.cap;
    hipStreamDestroy(c->stream);
    delete c;
    return fail(nullptr, "cannot allocate %zu MiB of workspace", cap >> 20);
  }
  c->stats_cap = (size_t)40 * clouds * 8 * kGnWords * 6;   // one registration's passes side by side: (2 + n_iter) P clouds for n_iter <= 10
  if (hipMalloc((void**)&c->stats, c->stats_cap * sizeof(double)) != hipSuccess) {
    hipFree(c->ws.base); hipStreamDestroy(c->stream);
    delete c;
    return fail(nullptr, "cannot allocate the statistics arena");
  }
  {
    // deep-level walker: device programs + two pinned staging sets + the tile queues / completion counters of one call's programs
    const size_t pb = sizeof(WalkProgram) * dsir_ctx::kWalkSlots;
    const size_t cb = sizeof(unsigned) * dsir_ctx::kWalkSlots * dsir_ctx::kWalkClouds * kWalkCtrWords;
    bool ok = hipMalloc((void**)&c->walk_dev, pb) == hipSuccess && hipMalloc((void**)&c->walk_ctr, cb) == hipSuccess &&
              hipMemset(c->walk_ctr, 0, cb) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
      ok = hipHostMalloc((void**)&c->walk_host[k], pb, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&c->walk_ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      for (int k = 0; k < 2; ++k) { if (c->walk_host[k]) hipHostFree(c->walk_host[k]); if (c->walk_ev[k]) hipEventDestroy(c->walk_ev[k]); }
      if (c->walk_dev) hipFree(c->walk_dev);
      if (c->walk_ctr) hipFree(c->walk_ctr);
      hipFree(c->stats); hipFree(c->ws.base); hipStreamDestroy(c->stream);
      delete c;
      return fail(nullptr, "cannot allocate the walker's program buffers");
    }
    c->walk_mode = tuning_flag("DSIR_WALK") ? 1 : 0;
    c->walk_wpc = (int)tuning_int("DSIR_WALK_WPC", 0);
    c->walk_flags = (int)tuning_int("DSIR_WALK_FLAGS", 0);
    if (tuning_flag("DSIR_WALK_TRACE")) {
      const size_t tb = sizeof(unsigned long long) * dsir_ctx::kWalkSlots * kWalkMaxPhases * 4;
      if (hipMalloc((void**)&c->walk_trace, tb) != hipSuccess) c->walk_trace = nullptr;
    }
  }
  if (hipMalloc((void**)&c->screen_acc, 6 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(c->screen_acc, 0, 6 * sizeof(unsigned long long)) != hipSuccess) {
    hipFree(c->stats); hipFree(c->ws.base); hipStreamDestroy(c->stream);
    delete c;
    return fail(nullptr, "cannot allocate the screening counters");
  }
  *out = c;
  return 0;
}

void dsir_destroy(dsir_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (auto& e : c->match_events) { hipEventDestroy(e.op0); hipEventDestroy(e.op1); hipEventDestroy(e.k0); hipEventDestroy(e.k1); }
  c->drop_graphs();
  if (c->dweights) hipFree(c->dweights);
  if (c->dweights16) hipFree(c->dweights16);
  if (c->match_ts) hipFree(c->match_ts);
  if (c->screen_acc) hipFree(c->screen_acc);
  for (int k = 0; k < 2; ++k) { if (c->walk_host[k]) hipHostFree(c->walk_host[k]); if (c->walk_ev[k]) hipEventDestroy(c->walk_ev[k]); }
  if (c->walk_dev) hipFree(c->walk_dev);
  if (c->walk_ctr) hipFree(c->walk_ctr);
  if (c->walk_trace) hipFree(c->walk_trace);
  if (c->stats) hipFree(c->stats);
  if (c->ws.base) hipFree(c->ws.base);
  for (int k = 0; k < dsir_ctx::kAux; ++k) if (c->aux[k]) { hipStreamSynchronize(c->aux[k]); hipStreamDestroy(c->aux[k]); }
  for (hipEvent_t e : c->fork_events) hipEventDestroy(e);
  hipStreamDestroy(c->own_stream);     // a caller's stream (dsir_set_stream) is the caller's to destroy
  delete c;
}

int dsir_gn_contributions(const dsir_cfg* cfg, int n_points) {
  if (!cfg || cfg->num_layers != 4 || n_points < 1) return -1;
  for (int l = 0; l < 4; ++l) if (cfg->sub_sampling_ratio[l] < 1) return -1;
  return gn_max_contributions(*cfg, n_points);
}
int dsir_gn_contribution_limit(void) { return kGnMaxContrib; }
int dsir_max_points_limit(const dsir_cfg* cfg) {
  if (dsir_gn_contributions(cfg, 1024) < 0) return -1;
  int lo = 1024, hi = 1 << 20;                       // the count grows with n: bisect the largest n within the bound
  if (gn_max_contributions(*cfg, hi) <= kGnMaxContrib) return hi;
  while (hi - lo > 1) {
    const int mid = lo + (hi - lo) / 2;
    if (gn_max_contributions(*cfg, mid) <= kGnMaxContrib) lo = mid; else hi = mid;
  }
  return lo;
}

void dsir_set_tuning(int on) { g_tuning = on ? 1 : 0; }
int dsir_tuning(void) { tuning_env("DSIR_TUNING"); return g_tuning == 1; }

const char* dsir_last_error(const dsir_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
void* dsir_stream(dsir_ctx* c) { return c ? (void*)c->stream : nullptr; }
int dsir_set_stream(dsir_ctx* c, void* stream, int restore_own) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  hipStream_t next = restore_own ? c->own_stream : (hipStream_t)stream;     // NULL is a valid caller stream: the legacy default stream
  if (next == c->stream) return 0;
  // Every call of a context re-uses its ONE workspace arena from the start: launches on the new stream must not overtake work
  // still running on the old one.  The new stream therefore waits (on the device, no host synchronisation) for everything
  // enqueued on the old stream so far.  Exception: a stream under capture cannot wait on outside work - whoever captures
  // synchronises before the capture begins (torch.cuda.graph does).
  hipStreamCaptureStatus cap_new = hipStreamCaptureStatusNone, cap_old = hipStreamCaptureStatusNone;
  hipStreamIsCapturing(next, &cap_new);
  hipStreamIsCapturing(c->stream, &cap_old);
  if (cap_new == hipStreamCaptureStatusNone && cap_old == hipStreamCaptureStatusNone) {
    hipEvent_t ev = nullptr;
    HIP_OK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(next, ev, 0);
    hipEventDestroy(ev);
    if (e != hipSuccess) return fail(c, "dsir_set_stream: ordering the new stream after the old one failed: %s", hipGetErrorString(e));
  }
  // a captured registration belongs to the old stream
  c->drop_graphs();
  c->stream = next;
  return 0;
}
int dsir_sync(dsir_ctx* c) {
  if (!c) return 1;
  HIP_OK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int dsir_num_weights(const dsir_ctx* c) { return c ? (int)c->params.size() : 0; }
const char* dsir_weight_name(const dsir_ctx* c, int i, int64_t* numel) {
  if (!c || i < 0 || i >= (int)c->params.size()) return nullptr;
  if (numel) *numel = c->params[i].ignored ? 1 : c->params[i].numel();
  return c->params[i].name.c_str();
}

int dsir_load_weight(dsir_ctx* c, const char* key, const float* host, const int64_t* shape, int ndim) {
  if (!c || !key) return 1;
  auto it = c->index.find(key);
  if (it == c->index.end()) return fail(c, "unexpected key in state_dict: %s", key);
  HostParam& p = c->params[it->second];
  if (p.ignored) { p.loaded = true; return 0; }
  if (!host) return fail(c, "null data for %s", key);
  if (ndim != (int)p.shape.size()) return fail(c, "size mismatch for %s: expected %d dims, got %d", key, (int)p.shape.size(), ndim);
  for (int d = 0; d < ndim; ++d)
    if (shape[d] != p.shape[d]) return fail(c, "size mismatch for %s: dim %d is %lld, expected %lld", key, d, (long long)shape[d], (long long)p.shape[d]);
  p.data.assign(host, host + p.numel());
  p.loaded = true;
  c->finalized = false;
  return 0;
}

int dsir_finalize_weights(dsir_ctx* c) {
  if (!c) return 1;
  for (auto& p : c->params)
    if (!p.loaded && !p.ignored) return fail(c, "missing key in state_dict: %s", p.name.c_str());
  Uploader u;
  const bool has_agg = c->cfg.pipeline != DSIR_PIPELINE_LABEL, has_inl = c->cfg.pipeline == DSIR_PIPELINE_ALIGN;
  RandlaOff fo = up_randla(c, u, "feat_extractor");
  RandlaOff io{};
  if (has_inl) io = up_randla(c, u, "inlier_model");
  LinOff mf[3] = {}, ma[5] = {}, mp{};
  if (has_agg) {
    mf[0] = up_lin(c, u, "mlp_feat", 0, true); mf[1] = up_lin(c, u, "mlp_feat", 3, true); mf[2] = up_lin(c, u, "mlp_feat", 6, false);
    ma[0] = up_lin(c, u, "mlp_att", 0, true); ma[1] = up_lin(c, u, "mlp_att", 3, true); ma[2] = up_lin(c, u, "mlp_att", 6, true);
    ma[3] = up_lin(c, u, "mlp_att", 9, true); ma[4] = up_lin(c, u, "mlp_att", 12, false);
    mp = up_lin(c, u, "mlp_proj", 0, false);
  }
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  // a captured registration holds the addresses of the old weight blob: drop it, the next call re-captures
  c->drop_graphs();
  if (c->dweights) { hipFree(c->dweights); c->dweights = nullptr; }
  HIP_OK(c, hipMalloc((void**)&c->dweights, u.blob.size() * sizeof(float)));
  HIP_OK(c, hipMemcpy(c->dweights, u.blob.data(), u.blob.size() * sizeof(float), hipMemcpyHostToDevice));
  const float* b = c->dweights;
  c->net.feat = bind_randla(b, fo, c->cfg);
  if (has_inl) c->net.inl = bind_randla(b, io, c->cfg);
  if (c->dweights16) { hipFree(c->dweights16); c->dweights16 = nullptr; }
  for (int k = 0; k < 5; ++k) c->agg_wh[k] = c->agg_wl[k] = nullptr;
  if (has_agg) {
    for (int k = 0; k < 3; ++k) c->net.mlp_feat[k] = bind_lin(b, mf[k]);
    for (int k = 0; k < 5; ++k) c->net.mlp_att[k] = bind_lin(b, ma[k]);
    c->net.mlp_proj = bind_lin(b, mp);
  }
  {
    // fp16 split (x -> fp16(x), fp16(x - fp16(x))) of the WHOLE blob (BatchNorm already folded), at the same offsets: the kernels
    // with an fp16-split contraction (agg_chain_h.hip, head_mlp_h.hip, pw_tile.hip) find the two parts of any matrix W at
    // dweights16 + (W - dweights) and dweights16 + nweights + (W - dweights).  20 MB for the align pipeline.
    u.blob.resize((u.blob.size() + 63) & ~(size_t)63, 0.f);     // the low parts start at dweights16 + total: keep them 16-byte aligned too
    const size_t total = u.blob.size();
    std::vector<uint16_t> h16(2 * total, 0);
    split_weights_f16(u.blob.data(), total, h16.data(), h16.data() + total);
    HIP_OK(c, hipMalloc((void**)&c->dweights16, h16.size() * sizeof(uint16_t)));
    HIP_OK(c, hipMemcpy(c->dweights16, h16.data(), h16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    c->nweights = total;
    auto hi = [&](size_t off) -> const void* { return c->dweights16 + off; };
    auto lo = [&](size_t off) -> const void* { return c->dweights16 + total + off; };
    if (has_agg) {
      const LinOff* lay[5] = {&ma[1], &ma[2], &ma[3], &ma[4], &mp};
      for (int k = 0; k < 5; ++k) { c->agg_wh[k] = hi(lay[k]->W); c->agg_wl[k] = lo(lay[k]->W); }
    }
    auto head = [&](const RandlaOff& o, RandlaW& w) {
      w.head_wh[0] = hi(o.out_w); w.head_wl[0] = lo(o.out_w);
      for (int k = 0; k < 3; ++k) { w.head_wh[k + 1] = hi(o.fc[k].W); w.head_wl[k + 1] = lo(o.fc[k].W); }
    };
    head(fo, c->net.feat);
    if (has_inl) head(io, c->net.inl);
  }
  c->finalized = true;
  return 0;
}

int dsir_narrow_i64(dsir_ctx* c, const int64_t* src, int32_t* dst, int64_t n) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  launch_narrow_i64(src, dst, n, c->stream);
  return post(c);
}

int dsir_knn_pyramid(dsir_ctx* c, const float* points, int stride, int clouds, int n, float* xyz, int32_t* neigh,
                     int32_t* sub, int32_t* interp) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  if (clouds < 1 || stride < 3) return fail(c, "dsir_knn_pyramid: bad arguments");
  c->ws.top = 0; c->ws.overflow = false;
  c->fork_events_used = 0;
  if (int r = build_pyramid(c, points, stride, clouds, n, xyz, neigh, sub, interp)) return r;
  return post(c);
}

int dsir_randla_forward(dsir_ctx* c, int which, const float* features, int cin, int clouds, int n, const float* xyz,
                        const int32_t* neigh, const int32_t* sub, const int32_t* interp, float* feat, float* logits) {
  if (check_ready(c)) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  if (which != 0 && c->cfg.pipeline != DSIR_PIPELINE_ALIGN) return fail(c, "randla_forward: this context has no inlier_model (pipeline != align)");
  const RandlaW& w = which == 0 ? c->net.feat : c->net.inl;
  if (cin != w.cin) return fail(c, "randla_forward: expected %d input channels, got %d", w.cin, cin);
  if (clouds > 2 * c->cfg.max_pairs || n > c->cfg.max_points) return fail(c, "randla_forward: batch exceeds max_pairs/max_points");
  Pyramid py;
  fill_pyramid_layout(c->cfg, clouds, n, py);
  if (py.nl[3] < kKnn) return fail(c, "cloud too small (n=%d)", n);
  py.xyz = xyz; py.neigh = neigh; py.sub = sub; py.interp = interp;
  const Seg in0 = plain_seg(features, (int64_t)n * cin, cin, cin);
  if (int r = walk_begin_call(c)) return r;
  if (int r = randla_forward(c, w, in0, nullptr, py, feat, logits)) return r;
  if (int r = walk_end_call(c)) return r;
  return post(c);
}

int dsir_score(dsir_ctx* c, const float* feat, const float* logits, const float* xyz, int64_t xyz_cs,
               const int32_t* neigh, int64_t neigh_cs, int clouds, int n, float* score, int32_t* label) {
  if (check_ready(c)) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  if (!feat || !logits || !xyz || !neigh || !score || clouds < 1 || n < 1) return fail(c, "dsir_score: bad arguments");
  if (clouds > 2 * c->cfg.max_pairs || n > c->cfg.max_points) return fail(c, "dsir_score: batch exceeds max_pairs/max_points");
  ScoreScratch s;
  s.red = c->ws.get<float>((size_t)clouds * 4);
  s.prob = c->ws.get<float>((size_t)clouds * n);
  s.label = c->ws.get<int32_t>((size_t)clouds * n);
  if (c->ws.overflow) return fail(c, "workspace exhausted in dsir_score");
  launch_score(feat, logits, c->cfg.num_classes, xyz, xyz_cs, neigh, neigh_cs, clouds, n, s, score, label, c->stream);
  return post(c);
}

int dsir_aggregate(dsir_ctx* c, const float* xyz, int64_t xyz_cs, const float* feat0, const float* score, int clouds,
                   int n, float* desc) {
  if (check_ready(c)) return 1;
  if (c->cfg.pipeline == DSIR_PIPELINE_LABEL) return fail(c, "dsir_aggregate: a label-pipeline context has no aggregation layers");
  HIP_OK(c, hipSetDevice(c->device));
  if (!xyz || !feat0 || !score || !desc || clouds < 1 || n < 1) return fail(c, "dsir_aggregate: bad arguments");
  // the workspace is sized from these two limits: inside them no allocation below can overflow
  if (clouds > 2 * c->cfg.max_pairs || n > c->cfg.max_points) return fail(c, "dsir_aggregate: batch exceeds max_pairs/max_points");
  float* F = run_mlp_feat(c, feat0, clouds, n);
  run_att_proj(c, xyz, xyz_cs, score, F, clouds, n, desc);
  return post(c);
}

int dsir_nn_match(dsir_ctx* c, const float* a, const float* b, int pairs, int J, int K, int32_t* idx) {
  if (!c) return 1;
  if (!a || !b || !idx || pairs < 1 || J < 1 || K < 1) return fail(c, "dsir_nn_match: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  void* scratch = c->ws.raw(nn_match_scratch_bytes(pairs, J, K));
  if (c->ws.overflow) return fail(c, "workspace exhausted in nn_match");
  dsir_ctx::MatchEvents* ev = match_event_slot(c);
  if (ev) hipEventRecord(ev->op0, c->stream);
  launch_nn_match_ws(a, b, pairs, J, K, idx, scratch, c->stream, ev ? ev->k0 : nullptr, ev ? ev->k1 : nullptr);
  if (ev) hipEventRecord(ev->op1, c->stream);
  return post(c);
}

int dsir_nn_match_screened(dsir_ctx* c, const float* a, const float* b, int pairs, int J, int K, int32_t* idx,
                           int64_t* stats) {
  if (!c) return 1;
  if (!a || !b || !idx || pairs < 1 || J < 1 || K < 1) return fail(c, "dsir_nn_match_screened: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  Arena& ws = c->ws;
  void* ah = ws.raw((size_t)pairs * J * 128); void* al = ws.raw((size_t)pairs * J * 128);
  void* bh = ws.raw((size_t)pairs * K * 128); void* bl = ws.raw((size_t)pairs * K * 128);
  float* sa = ws.get<float>((size_t)pairs * J); float* sb = ws.get<float>((size_t)pairs * K);
  void* scratch = ws.raw(nn_screen_scratch_bytes(pairs, J));
  unsigned long long* dstats = ws.get<unsigned long long>(2);
  int32_t* bad = ws.get<int32_t>(1);   // raised by the split when an element is outside the screening's domain
  if (ws.overflow) return fail(c, "workspace exhausted in nn_match_screened");
  hipStream_t st = c->stream;
  HIP_OK(c, hipMemsetAsync(bad, 0, 4, st));
  launch_split16_norm(a, (int64_t)pairs * J, ah, al, sa, st, bad);
  launch_split16_norm(b, (int64_t)pairs * K, bh, bl, sb, st, bad);
  launch_nn_screen(a, b, ah, al, bh, bl, sa, sb, pairs, J, K, idx, scratch, st, nullptr, nullptr, stats ? dstats : nullptr,
                   /*keep_gate=*/false, bad);
  if (stats) {
    HIP_OK(c, hipStreamSynchronize(st));
    unsigned long long h[2];
    HIP_OK(c, hipMemcpy(h, dstats, 16, hipMemcpyDeviceToHost));
    stats[0] = (int64_t)h[0]; stats[1] = (int64_t)h[1];
  }
  return post(c);
}

int dsir_screen_bounds(dsir_ctx* c, const float* a, const float* b, int J, int K, float* lower, float* upper, float* exact,
                       float* zacc, int32_t* idx, float* thresh, int32_t* cand_count, int32_t* cand_code, float* cand_lower,
                       int32_t* out_of_domain) {
  if (!c) return 1;
  if (!a || !b || !lower || !upper || !exact || !idx || !thresh || !cand_count || !cand_code || !cand_lower || J < 1 || K < 1 ||
      (int64_t)J * K > ((int64_t)1 << 26))
    return fail(c, "dsir_screen_bounds: bad arguments (J x K <= 2^26)");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  Arena& ws = c->ws;
  void* ah = ws.raw((size_t)J * 128); void* al = ws.raw((size_t)J * 128);
  void* bh = ws.raw((size_t)K * 128); void* bl = ws.raw((size_t)K * 128);
  float* sa = ws.get<float>((size_t)J); float* sb = ws.get<float>((size_t)K);
  void* scratch = ws.raw(nn_screen_scratch_bytes(1, J));
  int32_t* bad = ws.get<int32_t>(1);
  if (ws.overflow) return fail(c, "workspace exhausted in dsir_screen_bounds");
  hipStream_t st = c->stream;
  HIP_OK(c, hipMemsetAsync(bad, 0, 4, st));
  launch_split16_norm(a, J, ah, al, sa, st, bad);
  launch_split16_norm(b, K, bh, bl, sb, st, bad);
  launch_screen_bounds(a, b, ah, al, bh, bl, sa, sb, J, K, lower, upper, exact, zacc, st);
  // the product path on the same operands (bad == NULL: the screening runs even outside its domain, so that the flag
  // and the bound can be looked at independently), then its candidate lists
  launch_nn_screen(a, b, ah, al, bh, bl, sa, sb, 1, J, K, idx, scratch, st);
  launch_screen_export(scratch, J, thresh, cand_count, cand_code, cand_lower, st);
  if (out_of_domain) HIP_OK(c, hipMemcpyAsync(out_of_domain, bad, 4, hipMemcpyDeviceToDevice, st));
  return post(c);
}
int dsir_screen_cap(void) { return nn_screen_cap(); }

int dsir_kabsch(dsir_ctx* c, const float* src, const float* tgt, const float* w, int pairs, int m, float* T,
                int32_t* invalid) {
  if (!c) return 1;
  if (!src || !tgt || !w || !T || pairs < 1 || m < 1) return fail(c, "dsir_kabsch: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  if (invalid) HIP_OK(c, hipMemsetAsync(invalid, 0, sizeof(int32_t) * pairs, c->stream));
  KabschArgs a{};
  a.src = src; a.ref = tgt; a.idx = nullptr; a.w = w; a.src_stride = (int64_t)m * 3; a.ref_stride = (int64_t)m * 3;
  a.sigmoid = 0; a.pairs = pairs; a.m = m; a.T = T; a.invalid = invalid;
  a.chunk_min = c->kabsch_chunked_min;
  if (const size_t pb = kabsch_part_bytes(pairs, m, c->kabsch_chunked_min)) {      // large clouds: the chunked reduction of dsir_register (kabsch.hip)
    c->ws.top = 0; c->ws.overflow = false;
    a.part = c->ws.get<double>(pb / sizeof(double));
    if (c->ws.overflow) return fail(c, "dsir_kabsch: workspace exhausted");
  }
  launch_kabsch(a, c->stream);
  return post(c);
}

// Pyramids + forward_pair (model.py:609-648) of P pairs: everything both dsir_register and dsir_forward_pair need.
struct PairStage {
  Pyramid ps, pr;
  float *feat_s, *feat_r, *logit_s, *logit_r, *score_s, *score_r;
  int32_t *label_s, *label_r;     // only when want_label
  float *rxyz;                    // == pr.xyz
};
// pre: fills / copies the caller wants done before anything else; the stage adds its own (staging the input clouds, presetting the
// score reductions) and issues them all as ONE launch (launch_mem_ops)
static int forward_pair_stage(dsir_ctx* c, const dsir_pair_batch* in, bool want_score, bool want_label, PairStage& S,
                              int32_t* invalid = nullptr, MemOps pre = MemOps()) {
  const dsir_cfg& g = c->cfg;
  const int P = in->pairs, J = in->n_src, K = in->n_ref, cin = g.feat_len;
  if (P < 1 || P > g.max_pairs) return fail(c, "pairs=%d outside [1,%d]", P, g.max_pairs);
  if (J > g.max_points || K > g.max_points) return fail(c, "cloud larger than max_points=%d", g.max_points);
  if (!in->points_src || !in->points_ref) return fail(c, "null point clouds");
  const bool have_py = in->src_xyz && in->src_neigh && in->src_sub && in->src_interp && in->ref_xyz && in->ref_neigh &&
                       in->ref_sub && in->ref_interp;
  const bool any_py = in->src_xyz || in->src_neigh || in->src_sub || in->src_interp || in->ref_xyz || in->ref_neigh ||
                      in->ref_sub || in->ref_interp;
  if (any_py && !have_py) return fail(c, "either all eight pyramid tensors or none must be supplied");
  hipStream_t st = c->stream;
  Arena& ws = c->ws;

  // ---- pyramids of src and ref (engine-owned when built here)
  Pyramid& ps = S.ps; Pyramid& pr = S.pr;
  fill_pyramid_layout(g, P, J, ps);
  fill_pyramid_layout(g, P, K, pr);
  if (ps.nl[3] < kKnn || pr.nl[3] < kKnn) return fail(c, "cloud too small: need at least %d points", kKnn * 64);
  const bool joint = (J == K);   // src and ref share the feature extractor: run them as one batch of 2P clouds
  float* feat_all = ws.get<float>((size_t)P * (J + K) * 64);
  float* logit_all = ws.get<float>((size_t)P * (J + K) * g.num_classes);
  float* score_all = ws.get<float>((size_t)P * (J + K));
  int32_t* label_all = want_label ? ws.get<int32_t>((size_t)P * (J + K)) : nullptr;
  float* feat_s = feat_all; float* feat_r = feat_all + (size_t)P * J * 64;
  float* score_s = score_all; float* score_r = score_all + (size_t)P * J;
  float* logit_s = logit_all; float* logit_r = logit_all + (size_t)P * J * g.num_classes;
  int32_t* label_s = label_all; int32_t* label_r = label_all ? label_all + (size_t)P * J : nullptr;

  // pyramid storage: [src clouds | ref clouds] contiguous when joint
  float* pxyz = ws.get<float>((size_t)P * (ps.S + pr.S) * 3);
  int32_t* pneigh = ws.get<int32_t>((size_t)P * (ps.S + pr.S) * kKnn);
  int32_t* psub = ws.get<int32_t>((size_t)P * (ps.S1 + pr.S1) * kKnn);
  int32_t* pinterp = ws.get<int32_t>((size_t)P * (ps.S + pr.S));
  float* feats_in = ws.get<float>((size_t)P * (J + K) * cin);
  if (ws.overflow) return fail(c, "workspace exhausted (raise max_points / max_pairs)");
  float* rxyz = pxyz + (size_t)P * ps.S * 3;
  int32_t* rneigh = pneigh + (size_t)P * ps.S * kKnn;
  int32_t* rsub = psub + (size_t)P * ps.S1 * kKnn;
  int32_t* rinterp = pinterp + (size_t)P * ps.S;
  // the score stage's reduction targets (max feature, label weight, probability per cloud) live from here on: preset to -inf below
  float* score_red = want_score ? ws.get<float>((size_t)2 * P * 4) : nullptr;
  if (ws.overflow) return fail(c, "workspace exhausted (raise max_points / max_pairs)");
  pre.copy(feats_in, in->points_src, sizeof(float) * P * J * cin);
  pre.copy(feats_in + (size_t)P * J * cin, in->points_ref, sizeof(float) * P * K * cin);
  pre.fill(score_red, sizeof(float) * 2 * P * 4, 0xff800000u);
  if (have_py) {
    pre.copy(pxyz, in->src_xyz, sizeof(float) * P * ps.S * 3);
    pre.copy(rxyz, in->ref_xyz, sizeof(float) * P * pr.S * 3);
  }
  launch_mem_ops(pre, st);
  if (have_py) {
    // caller-supplied indices: copied with every entry clamped into its level's range (a bad index can never fault a
    // gather), out-of-range entries reported through bit 1 of the pair's invalid flag
    auto copy_idx = [&](const Pyramid& py, const int32_t* nb, const int32_t* sb, const int32_t* ip, int32_t* nbo, int32_t* sbo,
                        int32_t* ipo) {
      PyramidIdxCopy a{};
      a.neigh = nb; a.sub = sb; a.interp = ip; a.neigh_out = nbo; a.sub_out = sbo; a.interp_out = ipo;
      a.S = py.S; a.S1 = py.S1; a.levels = g.num_layers;
      for (int l = 0; l <= g.num_layers; ++l) { a.nl[l] = py.nl[l]; a.off[l] = py.off[l]; a.soff[l] = py.soff[l]; }
      a.flag = invalid; a.flag_mod = P;
      launch_copy_pyramid_idx(a, P, st);
    };
    copy_idx(ps, in->src_neigh, in->src_sub, in->src_interp, pneigh, psub, pinterp);
    copy_idx(pr, in->ref_neigh, in->ref_sub, in->ref_interp, rneigh, rsub, rinterp);
  } else if (joint) {
    if (int r = build_pyramid(c, feats_in, cin, 2 * P, J, pxyz, pneigh, psub, pinterp)) return r;
  } else {
    if (int r = build_pyramid(c, feats_in, cin, P, J, pxyz, pneigh, psub, pinterp)) return r;
    if (int r = build_pyramid(c, feats_in + (size_t)P * J * cin, cin, P, K, rxyz, rneigh, rsub, rinterp)) return r;
  }
  ps.xyz = pxyz; ps.neigh = pneigh; ps.sub = psub; ps.interp = pinterp;
  pr.xyz = rxyz; pr.neigh = rneigh; pr.sub = rsub; pr.interp = rinterp;

  // ---- forward_pair (model.py:609-648): feature RandLA + score on src and ref
  const size_t mark0 = ws.mark();
  ScoreScratch sc;
  if (joint) {
    Pyramid pa = ps;
    pa.clouds = 2 * P;
    if (int r = randla_forward(c, c->net.feat, plain_seg(feats_in, (int64_t)J * cin, cin, cin), nullptr, pa, feat_all, logit_all)) return r;
    if (want_score) {
      sc.red = score_red; sc.prob = ws.get<float>((size_t)2 * P * J); sc.label = ws.get<int32_t>((size_t)2 * P * J);
      launch_score(feat_all, logit_all, g.num_classes, pxyz, (int64_t)ps.S * 3, pneigh, (int64_t)ps.S * kKnn, 2 * P, J, sc,
                   score_all, label_all, st, /*red_preset=*/true);
    }
  } else {
    if (int r = randla_forward(c, c->net.feat, plain_seg(feats_in, (int64_t)J * cin, cin, cin), nullptr, ps, feat_s, logit_s)) return r;
    ws.release(mark0);
    if (int r = randla_forward(c, c->net.feat, plain_seg(feats_in + (size_t)P * J * cin, (int64_t)K * cin, cin, cin), nullptr, pr, feat_r, logit_r)) return r;
    ws.release(mark0);
    if (want_score) {
      const int nmax = J > K ? J : K;
      sc.red = score_red; sc.prob = ws.get<float>((size_t)P * nmax); sc.label = ws.get<int32_t>((size_t)P * nmax);
      launch_score(feat_s, logit_s, g.num_classes, pxyz, (int64_t)ps.S * 3, pneigh, (int64_t)ps.S * kKnn, P, J, sc, score_s, label_s, st, true);
      sc.red = score_red + (size_t)P * 4;      // the ref clouds' own targets (the src launch is still reading the first set)
      launch_score(feat_r, logit_r, g.num_classes, rxyz, (int64_t)pr.S * 3, rneigh, (int64_t)pr.S * kKnn, P, K, sc, score_r, label_r, st, true);
    }
  }
  ws.release(mark0);
  S.feat_s = feat_s; S.feat_r = feat_r; S.logit_s = logit_s; S.logit_r = logit_r; S.score_s = score_s; S.score_r = score_r;
  S.label_s = label_s; S.label_r = label_r; S.rxyz = rxyz;
  return 0;
}

static unsigned long long* match_ts_slot(dsir_ctx* c) {
  if (!c->time_match || !c->match_ts || c->match_ts_used >= kMatchSlots) return nullptr;
  return c->match_ts + 2 * c->match_ts_used++;
}

static int register_enqueue(dsir_ctx* c, const dsir_pair_batch* in, int n_iter, const dsir_pair_result* out) {
  const dsir_cfg& g = c->cfg;
  const int P = in->pairs, J = in->n_src, K = in->n_ref;
  if (n_iter < 1) return fail(c, "n_iter must be >= 1");
  hipStream_t st = c->stream;
  Arena& ws = c->ws;
  PairStage S;
  // What opens a registration goes out as ONE launch (launch_mem_ops, issued by forward_pair_stage together with the staging of the
  // input clouds): the pairs' flags, and the GroupNorm statistics of ALL passes of this call (feature extractor on 2 P clouds, n_iter
  // inlier passes on P) plus those of the inlier model's cached position-encoding branch (EncCache) in one zeroed region - round 4
  // spent six memset / memcpy launches here
  MemOps pre;
  pre.fill(out->invalid, sizeof(int32_t) * P);
  if (int r = walk_begin_call(c)) return r;
  struct StatsGuard { dsir_ctx* c; ~StatsGuard() { c->stats_prezeroed = false; c->stats_base = 0; } } stats_guard{c};
  const size_t cache_stats = (size_t)2 * g.num_layers * P * 8 * kGnWords;      // EncCache: two layers per level
  double* cache_stats_at = nullptr;
  {
    const size_t total = (size_t)40 * 8 * kGnWords * ((size_t)2 * P + (size_t)n_iter * P);
    if (total + cache_stats <= c->stats_cap) {
      pre.fill(c->stats, (total + cache_stats) * sizeof(double));
      // ... and the tile queues / completion counters of the deep-level walker's programs (walk.hip), one per pass
      if (c->walk_mode && c->walk_ctr && P <= dsir_ctx::kWalkClouds)
        pre.fill(c->walk_ctr, sizeof(unsigned) * dsir_ctx::kWalkSlots * dsir_ctx::kWalkClouds * kWalkCtrWords);
      cache_stats_at = c->stats + total;
      c->stats_prezeroed = true;
      c->stats_base = 0;
    }
  }
  if (int r = forward_pair_stage(c, in, true, false, S, out->invalid, pre)) return r;
  const Pyramid& ps = S.ps; const Pyramid& pr = S.pr;
  float *feat_s = S.feat_s, *feat_r = S.feat_r, *score_s = S.score_s, *score_r = S.score_r, *rxyz = S.rxyz;
  const float* pxyz = ps.xyz;

  // ---- loop invariants of Network.aggregation: the whole ref side and mlp_feat(feat_src)
  float* desc_r = ws.get<float>((size_t)P * K * 64);
  float* desc_s = ws.get<float>((size_t)P * J * 64);
  float* F_s = ws.get<float>((size_t)P * J * 64);
  float* xyz_cur = ws.get<float>((size_t)P * J * 3);
  float* logits_it = ws.get<float>((size_t)P * J);
  int32_t* idx_it = ws.get<int32_t>((size_t)P * J);
  float* T_it = ws.get<float>((size_t)P * 12);
  void* match_scratch = ws.raw(nn_match_scratch_bytes(P, J, K));
  // fp16-screened arg-min (nn_screen.hip): split descriptors, norms, candidate scratch.  The ref side is loop invariant.
  // both paths return the same bits, so the choice is free: small problems (latency-bound, e.g. one pair in flight) take
  // the single exhaustive kernel, large ones the three-kernel screened path.  dsir_enable_screen / DSIR_NO_SCREEN: A/B
  // switch to the exhaustive fp32 kernel throughout
  static const long long screen_min = tuning_int("DSIR_SCREEN_MIN_WORK", 100000000ll);   // A/B hook
  const bool screen = c->screen_mode && !in->forced_idx && (int64_t)P * J * K >= screen_min;
  void *sc_ah = nullptr, *sc_al = nullptr, *sc_bh = nullptr, *sc_bl = nullptr, *sc_scratch = nullptr;
  float *sc_sa = nullptr, *sc_sb = nullptr;
  if (screen) {
    sc_ah = ws.raw((size_t)P * J * 64 * 2); sc_al = ws.raw((size_t)P * J * 64 * 2);
    sc_bh = ws.raw((size_t)P * K * 64 * 2); sc_bl = ws.raw((size_t)P * K * 64 * 2);
    sc_sa = ws.get<float>((size_t)P * J); sc_sb = ws.get<float>((size_t)P * K);
    sc_scratch = ws.raw(nn_screen_scratch_bytes(P, J));
  }
  // pruned search (nn_prune.hip) for long ref ranges: column order + tile bounds once, row order + tile lists per iteration
  // ... and only with enough rows in the launch to fill the chip with items (128 row blocks): below that the search lasts as long as
  // its longest item either way and the preparation is pure cost (one 16384-point pair: 5.18 -> 5.59 ms per registration with it)
  const bool prune = screen && c->prune_min_points > 0 && K >= c->prune_min_points && (int64_t)P * J >= c->prune_min_rows &&
                     nn_prune_supported(P, J, K);
  void* pr_scratch = prune ? ws.raw(nn_prune_scratch_bytes(P, J, K)) : nullptr;
  // chunk partials of the pose solve on large clouds (kabsch.hip)
  const size_t kab_bytes = kabsch_part_bytes(P, J, c->kabsch_chunked_min);
  double* kab_part = kab_bytes ? ws.get<double>(kab_bytes / sizeof(double)) : nullptr;
  // persistent storage of the inlier model's position-encoding branch (EncCache), alive across the iterations
  EncCache enc_cache;
  static const bool no_hoist = tuning_flag("DSIR_NO_HOIST");   // A/B switch
  const bool hoist = !no_hoist && n_iter > 1;
  if (hoist) {
    size_t nstats = 0;
    for (int l = 0; l < g.num_layers; ++l) {
      const size_t rows = (size_t)P * ps.nl[l] * kKnn, ch = (size_t)g.d_out[l] / 2;
      if (c->net.inl.blk[l].lse_w8 && lse_uv_enabled()) {
        enc_cache.uv_buf[l] = ws.get<float>((size_t)P * ps.nl[l] * 2 * ch);
        enc_cache.dist_buf[l] = ws.get<float>(rows);
      } else {
        enc_cache.enc_buf[l] = ws.get<float>(rows * ch);
      }
      enc_cache.enc2_buf[l] = ws.get<float>(rows * ch);
      static const bool no_s2 = tuning_flag("DSIR_NO_S2");   // A/B switch: recompute the enc half of the scores every iteration
      // level 1 (d = 64: a 32-channel contraction) caches its score halves only for a few pairs in flight: with the chip full
      // re-reading 64 floats per row costs more HBM time than contracting 32 (same bits either way; +1.9 % pairs/s at 128
      // pairs per launch, -0.02 ms of single-pair latency with the cache)
      static const int s2_min_d = (int)tuning_int("DSIR_S2_MIN_D", 0);   // tuning hook: 0 = by launch size
      // round 3: with the score contraction on the fp16 pipe, re-reading level 2's halves (2 x 5000 x 128 floats per cloud) also
      // costs more than contracting them when the chip is full: only level 3 keeps its cache there (+0.7 % pairs/s; 64: -0.7 %)
      const int min_d = s2_min_d > 0 ? s2_min_d : ((P <= 4 && !att_pool_enabled()) ? 64 : 256);   // att_pool.hip (d = 64, 128) keeps no score cache
      if (g.d_out[l] >= 64 && g.d_out[l] >= min_d && !no_s2) {
        enc_cache.s2_buf[l][0] = ws.get<float>(rows * (size_t)g.d_out[l]);
        enc_cache.s2_buf[l][1] = ws.get<float>(rows * (size_t)g.d_out[l]);
      }
      nstats += 2 * (size_t)P * 8 * kGnWords;
    }
    double* cst = cache_stats_at;                  // zeroed by the opening launch
    if (!cst || nstats > cache_stats) {            // more iterations than the statistics arena holds side by side: own storage, own memset
      cst = ws.get<double>(nstats);
      if (!ws.overflow) HIP_OK(c, hipMemsetAsync(cst, 0, nstats * sizeof(double), st));
    }
    for (int l = 0; l < g.num_layers; ++l) {
      enc_cache.enc_stats[l] = cst + (size_t)(2 * l) * P * 8 * kGnWords;
      enc_cache.enc2_stats[l] = cst + (size_t)(2 * l + 1) * P * 8 * kGnWords;
    }
  }
  if (ws.overflow) return fail(c, "workspace exhausted (raise max_points / max_pairs)");
  const size_t mark1 = ws.mark();
  {
    // the two loop-invariant halves are independent of each other: with few clouds in flight mlp_feat(feat_src) (and the copy of the
    // src coordinates) runs on an auxiliary stream beside the ref side's chain; its temporaries keep their own arena space until the join
    const bool forked = fork_on(c, 2 * P, 8);
    hipStream_t side = forked ? c->aux[0] : st;
    if (forked) fork_to(c, st, side);
    c->stream = side;                         // run_mlp_feat launches on the context's stream
    run_mlp_feat(c, feat_s, P, J, F_s);       // straight into the storage that outlives the iterations
    c->stream = st;
    launch_copy_xyz(pxyz, (int64_t)ps.S * 3, 3, J, P, xyz_cur, (int64_t)J * 3, side);      // xyz_cur = level-0 src coordinates
    float* F_r = run_mlp_feat(c, feat_r, P, K);
    AggExtras exr;
    if (screen) { exr.sq = sc_sb; exr.hi = sc_bh; exr.lo = sc_bl; }
    const bool ref_prepared = run_att_proj(c, rxyz, (int64_t)pr.S * 3, score_r, F_r, P, K, desc_r, screen ? &exr : nullptr);
    if (out->desc_ref) HIP_OK(c, hipMemcpyAsync(out->desc_ref, desc_r, sizeof(float) * P * K * 64, hipMemcpyDeviceToDevice, st));
    if (screen && !ref_prepared) {
      launch_split16_norm(desc_r, (int64_t)P * K, sc_bh, sc_bl, sc_sb, st);
    }
    if (prune && launch_prune_ref(rxyz, (int64_t)pr.S * 3, desc_r, sc_bh, sc_bl, sc_sb, P, J, K, pr_scratch, st)) return fail(c, "pruned search: sorting the ref side failed");
    if (forked) fork_to(c, side, st);
    ws.release(mark1);
  }

  for (int it = 0; it < n_iter; ++it) {
    int32_t* idx_out = out->idx ? out->idx + (size_t)it * P * J : idx_it;
    float* logit_out = out->logits ? out->logits + (size_t)it * P * J : logits_it;
    // aggregation of the (transformed) src cloud
    // aggregation of the (transformed) src cloud; its epilogue also leaves what this iteration's search needs of the descriptors
    // (screened search: the fp16 operand pair + norms; exhaustive search: norms + preset result slots)
    AggExtras exs;
    if (!in->forced_idx) {
      if (screen) { exs.sq = sc_sa; exs.hi = sc_ah; exs.lo = sc_al; }
      else nn_match_scratch_layout(match_scratch, P, J, K, &exs.sq, &exs.packed_init);
    }
    const bool src_prepared = run_att_proj(c, xyz_cur, (int64_t)J * 3, score_s, F_s, P, J, desc_s, in->forced_idx ? nullptr : &exs);
    ws.release(mark1);
    if (out->desc_src)
      HIP_OK(c, hipMemcpyAsync(out->desc_src + (size_t)it * P * J * 64, desc_s, sizeof(float) * P * J * 64, hipMemcpyDeviceToDevice, st));
    // nearest ref descriptor
    if (in->forced_idx) {
      // caller-supplied correspondences: clamped into [0, K), out-of-range entries reported through the pair's flag
      launch_copy_idx_clamped(in->forced_idx + (size_t)it * P * J, J, J, K, P, idx_out, J, out->invalid, P, st);
    } else {
      // HIP events on the engine's stream: op0..op1 around every kernel of the operation (split, screening, pick,
      // fallback / norms, search, unpack), k0..k1 around its dominant kernel alone
      dsir_ctx::MatchEvents* ev = match_event_slot(c);
      if (ev) hipEventRecord(ev->op0, st);
      if (screen) {
        if (!src_prepared) launch_split16_norm(desc_s, (int64_t)P * J, sc_ah, sc_al, sc_sa, st);
        ScreenOrder ord;
        if (prune) {
          // an actual distance of every row - to its previous match, to the columns of its nearest tile - bounds its minimum from
          // above: skip the tiles that cannot beat it (iteration 0 has only the second kind)
          const int32_t* idx_prev = it == 0 ? nullptr : (out->idx ? out->idx + (size_t)(it - 1) * P * J : idx_it);
          if (launch_prune_rows(desc_s, desc_r, sc_ah, sc_al, sc_sa, sc_sb, idx_prev, P, J, K, pr_scratch, st, &ord, c->screen_acc + 4))
            return fail(c, "pruned search: sorting the rows failed");
        }
        launch_nn_screen(desc_s, desc_r, sc_ah, sc_al, sc_bh, sc_bl, sc_sa, sc_sb, P, J, K, idx_out, sc_scratch, st, nullptr, nullptr, nullptr,
                         /*keep_gate=*/it > 0, nullptr, c->screen_acc, ev ? ev->k0 : nullptr, ev ? ev->k1 : nullptr, ord);
      } else {
        ++c->exhaustive_searches;
        launch_nn_match_ws(desc_s, desc_r, P, J, K, idx_out, match_scratch, st, ev ? ev->k0 : nullptr, ev ? ev->k1 : nullptr,
                           /*ref_norms_cached=*/it > 0, match_ts_slot(c), /*src_norms_ready=*/src_prepared);
      }
      if (ev) hipEventRecord(ev->op1, st);
    }
    // inlier RandLA on [xyz_src(t); xyz_ref[idx]] with the SRC pyramid (model.py:574-577)
    const Seg s0 = plain_seg(xyz_cur, (int64_t)J * 3, 3, 3);
    const Seg s1 = plain_seg(rxyz, (int64_t)pr.S * 3, 3, 3, idx_out, J);
    if (int r = randla_forward(c, c->net.inl, s0, &s1, ps, nullptr, logit_out, hoist ? &enc_cache : nullptr)) return r;
    ws.release(mark1);
    // weighted Kabsch + transform update (model.py:586-595)
    KabschArgs a{};
    a.src = xyz_cur; a.ref = rxyz; a.idx = idx_out; a.w = logit_out; a.src_stride = (int64_t)J * 3; a.ref_stride = (int64_t)pr.S * 3;
    a.sigmoid = 1; a.pairs = P; a.m = J; a.T = T_it; a.invalid = out->invalid;
    a.src_out = xyz_cur; a.src_out_stride = (int64_t)J * 3;
    a.T_cum = out->transforms + (size_t)it * 12; a.T_prev = it ? out->transforms + (size_t)(it - 1) * 12 : nullptr;
    a.T_stride = (int64_t)n_iter * 12;
    a.matched_out = (it == n_iter - 1) ? out->pt_ref_new : nullptr;
    a.part = kab_part; a.chunk_min = c->kabsch_chunked_min;
    launch_kabsch(a, st);
  }
  if (c->ws.overflow) return fail(c, "workspace exhausted (raise max_points / max_pairs in dsir_cfg)");
  if (int r = walk_end_call(c)) return r;
  return 0;
}

int dsir_register(dsir_ctx* c, const dsir_pair_batch* in, int n_iter, const dsir_pair_result* out) {
  if (check_ready(c)) return 1;
  if (!in || !out || !out->transforms) return fail(c, "dsir_register: null argument");
  if (c->cfg.pipeline != DSIR_PIPELINE_ALIGN) return fail(c, "dsir_register needs an align-pipeline context");
  HIP_OK(c, hipSetDevice(c->device));
  if (!c->use_graph || c->time_match) {
    if (int r = register_enqueue(c, in, n_iter, out)) return r;
    return post(c);
  }
  // Graph mode: the whole launch sequence (~450 kernels) is captured once per distinct call
  // signature (sizes AND buffer addresses) and replayed with one hipGraphLaunch.
  std::vector<unsigned char> key(sizeof(*in) + sizeof(*out) + sizeof(int));
  std::memcpy(key.data(), in, sizeof(*in));
  std::memcpy(key.data() + sizeof(*in), out, sizeof(*out));
  std::memcpy(key.data() + sizeof(*in) + sizeof(*out), &n_iter, sizeof(int));
  hipGraphExec_t exec = nullptr;
  for (auto& g : c->graphs)
    if (g.key == key) { exec = g.exec; break; }
  if (!exec) {
    HIP_OK(c, hipStreamSynchronize(c->stream));
    // the walker programs of this registration get a device block of their own (the kernel nodes hold addresses inside it); they are
    // collected on the host while the launches are captured and uploaded once, below
    void* walk_block = nullptr;
    if (c->walk_mode && c->walk_dev) {
      HIP_OK(c, hipMalloc(&walk_block, sizeof(WalkProgram) * dsir_ctx::kWalkSlots));
      c->cap_host.assign(sizeof(WalkProgram) * dsir_ctx::kWalkSlots, 0);
      c->cap_dev = reinterpret_cast<WalkProgram*>(walk_block);
    }
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      if (walk_block) hipFree(walk_block);
      return fail(c, "hipStreamBeginCapture failed");
    }
    c->capturing = true;
    const int rc = register_enqueue(c, in, n_iter, out);
    c->capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != 0) { if (graph) hipGraphDestroy(graph); if (walk_block) hipFree(walk_block); return rc; }
    if (e != hipSuccess || !graph) { if (walk_block) hipFree(walk_block); return fail(c, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
    if (walk_block && c->walk_used > 0)
      HIP_OK(c, hipMemcpy(walk_block, c->cap_host.data(), sizeof(WalkProgram) * (size_t)c->walk_used, hipMemcpyHostToDevice));
    c->cap_dev = nullptr;
    {
      // what one registration costs in launches: the node census of the captured graph (dsir_graph_stats)
      size_t nn = 0;
      c->graph_nodes[0] = c->graph_nodes[1] = c->graph_nodes[2] = c->graph_nodes[3] = 0;
      if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess && nn > 0) {
        std::vector<hipGraphNode_t> nodes(nn);
        if (hipGraphGetNodes(graph, nodes.data(), &nn) == hipSuccess) {
          c->graph_nodes[0] = (int64_t)nn;
          for (size_t i = 0; i < nn; ++i) {
            hipGraphNodeType t;
            if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) continue;
            if (t == hipGraphNodeTypeKernel) ++c->graph_nodes[1];
            else if (t == hipGraphNodeTypeMemset) ++c->graph_nodes[2];
            else if (t == hipGraphNodeTypeMemcpy) ++c->graph_nodes[3];
          }
        }
      }
    }
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e2 != hipSuccess) { if (walk_block) hipFree(walk_block); return fail(c, "hipGraphInstantiate: %s", hipGetErrorString(e2)); }
    if (c->graphs.size() >= dsir_ctx::kMaxGraphs) {
      HIP_OK(c, hipStreamSynchronize(c->stream));     // the evicted graph may still be replaying
      hipGraphExecDestroy(c->graphs.front().exec);
      if (c->graphs.front().walk_block) hipFree(c->graphs.front().walk_block);
      c->graphs.erase(c->graphs.begin());
    }
    c->graphs.push_back({key, exec, walk_block});
  }
  HIP_OK(c, hipGraphLaunch(exec, c->stream));
  return post(c);
}

// One side of forward_pair's endpoints (model.py:637-666).
static int emit_cloud_out(dsir_ctx* c, const Pyramid& py, const float* feat0, const float* logits, const float* score,
                          const int32_t* label, int P, int n, int num_sub, const dsir_cloud_out* o) {
  const dsir_cfg& g = c->cfg;
  hipStream_t st = c->stream;
  Arena& ws = c->ws;
  const int M = num_sub > 0 ? num_sub : n;
  const int64_t xyz_cs = (int64_t)py.S * 3;
  if (o->logits) HIP_OK(c, hipMemcpyAsync(o->logits, logits, sizeof(float) * P * n * g.num_classes, hipMemcpyDeviceToDevice, st));
  if (g.pipeline == DSIR_PIPELINE_LABEL) {
    if (o->xyz) launch_copy_xyz(py.xyz, xyz_cs, 3, n, P, o->xyz, (int64_t)n * 3, st);
    if (o->feat) launch_l2norm64(feat0, (int64_t)P * n, o->feat, st);
    return 0;
  }
  const size_t mark = ws.mark();
  const int32_t* sel = nullptr;
  const float* xyz_m = py.xyz; int64_t xyz_m_cs = xyz_cs;
  const float* feat_m = feat0; const float* score_m = score;
  if (num_sub > 0) {
    int32_t* idx = ws.get<int32_t>((size_t)P * M);
    float* sc = ws.get<float>((size_t)P * M);
    float* xs = ws.get<float>((size_t)P * M * 3);
    float* fs = ws.get<float>((size_t)P * M * 64);
    void* scratch = ws.raw(topk_scratch_bytes(P, n));
    if (ws.overflow) return fail(c, "workspace exhausted in forward_pair");
    if (int r = launch_topk(score, P, n, M, idx, sc, scratch, st)) return fail(c, "top-k selection failed (%d)", r);
    launch_gather_rows(py.xyz, xyz_cs, 3, idx, 3, M, P, xs, st);
    launch_gather_rows(feat0, (int64_t)n * 64, 64, idx, 64, M, P, fs, st);
    sel = idx; xyz_m = xs; xyz_m_cs = (int64_t)M * 3; feat_m = fs; score_m = sc;
  }
  if (o->xyz) {
    if (sel) HIP_OK(c, hipMemcpyAsync(o->xyz, xyz_m, sizeof(float) * P * M * 3, hipMemcpyDeviceToDevice, st));
    else launch_copy_xyz(py.xyz, xyz_cs, 3, n, P, o->xyz, (int64_t)n * 3, st);
  }
  if (o->score) HIP_OK(c, hipMemcpyAsync(o->score, score_m, sizeof(float) * P * M, hipMemcpyDeviceToDevice, st));
  if (o->label) launch_gather_i32(label, n, sel, M, P, o->label, st);
  if (o->index && sel) HIP_OK(c, hipMemcpyAsync(o->index, sel, sizeof(int32_t) * P * M, hipMemcpyDeviceToDevice, st));
  if (o->feat) {
    if (g.pipeline == DSIR_PIPELINE_ALIGN) {
      HIP_OK(c, hipMemcpyAsync(o->feat, feat_m, sizeof(float) * P * M * 64, hipMemcpyDeviceToDevice, st));
    } else {
      // aggregation (already L2-normalised, model.py:232-233), then the second F.normalize of :650-651
      float* desc = ws.get<float>((size_t)P * M * 64);
      if (ws.overflow) return fail(c, "workspace exhausted in forward_pair");
      const size_t m2 = ws.mark();
      float* F = run_mlp_feat(c, feat_m, P, M);
      run_att_proj(c, xyz_m, xyz_m_cs, score_m, F, P, M, desc);
      ws.release(m2);
      launch_l2norm64(desc, (int64_t)P * M, o->feat, st);
    }
  }
  ws.release(mark);
  return 0;
}

int dsir_forward_pair(dsir_ctx* c, const dsir_pair_batch* in, int num_sub, const dsir_cloud_out* src, const dsir_cloud_out* ref) {
  if (check_ready(c)) return 1;
  if (!in || !src || !ref) return fail(c, "dsir_forward_pair: null argument");
  HIP_OK(c, hipSetDevice(c->device));
  const dsir_cfg& g = c->cfg;
  if (num_sub > 0) {
    if (g.pipeline != DSIR_PIPELINE_FEAT) return fail(c, "num_sub > 0 is only meaningful for the feat pipeline");
    if (num_sub > in->n_src || num_sub > in->n_ref) return fail(c, "num_sub=%d exceeds the cloud size", num_sub);
  }
  const bool scored = g.pipeline != DSIR_PIPELINE_LABEL;
  PairStage S;
  if (int r = walk_begin_call(c)) return r;
  if (int r = forward_pair_stage(c, in, scored, scored, S)) return r;
  if (int r = walk_end_call(c)) return r;
  if (int r = emit_cloud_out(c, S.ps, S.feat_s, S.logit_s, S.score_s, S.label_s, in->pairs, in->n_src, num_sub, src)) return r;
  if (int r = emit_cloud_out(c, S.pr, S.feat_r, S.logit_r, S.score_r, S.label_r, in->pairs, in->n_ref, num_sub, ref)) return r;
  return post(c);
}

int dsir_icp_refine(dsir_ctx* c, const float* points_src, const float* points_ref, int pairs, int J, int K, int stride,
                    float max_corr_dist, int max_iter, float rel_fitness, float rel_rmse, const float* T_init,
                    float* T_out, double* stats) {
  if (!c) return 1;
  if (!points_src || !points_ref || !T_init || !T_out || pairs < 1 || J < 1 || K < 1 || stride < 3 || max_iter < 0 ||
      !(max_corr_dist > 0.f))
    return fail(c, "dsir_icp_refine: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  void* scratch = c->ws.raw(icp_scratch_bytes(pairs, J));
  if (c->ws.overflow) return fail(c, "workspace too small for dsir_icp_refine (raise max_points / max_pairs)");
  launch_icp_refine(points_src, points_ref, pairs, J, K, stride, max_corr_dist, max_iter, rel_fitness, rel_rmse, T_init,
                    T_out, stats, scratch, c->stream);
  return post(c);
}

int dsir_pose_finetune(dsir_ctx* c, const float* xyz_src, const float* xyz_ref, const float* weights, int weights_are_logits,
                       int pairs, int m, const float* T_init, float quantization_size, int max_iter, float break_threshold_ratio,
                       int max_break_count, float* T_out, double* stats) {
  if (!c) return 1;
  if (!xyz_src || !xyz_ref || !T_init || !T_out || pairs < 1 || m < 1 || max_iter < 0 || max_break_count < 1 ||
      !(quantization_size > 0.f) || !(break_threshold_ratio >= 0.f))
    return fail(c, "dsir_pose_finetune: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  launch_pose_finetune(xyz_src, xyz_ref, weights, weights_are_logits ? 1 : 0, pairs, m, T_init, quantization_size, max_iter,
                       break_threshold_ratio, max_break_count, T_out, stats, c->stream);
  return post(c);
}

int dsir_align_loss_backward2(dsir_ctx* c, const float* pt_src, const float* pt_ref, const int32_t* idx, const float* logits,
                             const float* labels, const float* transform_gt, int pairs, int J, int K, int n_iter, int loss_type,
                             float wt_ptDist_loss, float wt_inlier_loss, float loss_discount_factor, float* transforms,
                             double* losses, float* grad_logits, double* losses_per_pair) {
  if (!c) return 1;
  if (!pt_src || !pt_ref || !idx || !logits || !transform_gt || !grad_logits || pairs < 1 || J < 1 || K < 1 || n_iter < 1 ||
      n_iter > 8 || (loss_type != 0 && loss_type != 1))
    return fail(c, "dsir_align_loss_backward: bad arguments (n_iter in [1,8], loss_type 0 = mae / 1 = mse)");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  // correspondences come from the caller: clamped into [0, K) before any gather
  int32_t* idx_ok = c->ws.get<int32_t>((size_t)n_iter * pairs * J);
  double* dloss = c->ws.get<double>((size_t)2 * n_iter);
  double* dpart = c->ws.get<double>((size_t)2 * n_iter * pairs);       // every pair's loss terms, added in pair order (align_loss.hip)
  if (c->ws.overflow) return fail(c, "workspace too small for dsir_align_loss_backward");
  launch_copy_idx_clamped(idx, (int64_t)pairs * J, pairs * J, K, n_iter, idx_ok, (int64_t)pairs * J, nullptr, 1, c->stream);
  if (launch_align_loss(pt_src, pt_ref, idx_ok, logits, labels, transform_gt, pairs, J, K, n_iter, loss_type, wt_ptDist_loss,
                        wt_inlier_loss, loss_discount_factor, transforms, dloss, grad_logits, c->stream, dpart))
    return fail(c, "dsir_align_loss_backward: launch failed");
  if (losses || losses_per_pair) {
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (losses) HIP_OK(c, hipMemcpy(losses, dloss, sizeof(double) * 2 * n_iter, hipMemcpyDeviceToHost));
    if (losses_per_pair) {
      // the kernel's per-pair partials carry the batch mean's 1 / pairs: a pair's own mean is pairs x its share
      HIP_OK(c, hipMemcpy(losses_per_pair, dpart, sizeof(double) * 2 * n_iter * pairs, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < (size_t)2 * n_iter * pairs; ++k) losses_per_pair[k] *= (double)pairs;
    }
  }
  return post(c);
}

int dsir_align_loss_backward(dsir_ctx* c, const float* pt_src, const float* pt_ref, const int32_t* idx, const float* logits,
                             const float* labels, const float* transform_gt, int pairs, int J, int K, int n_iter, int loss_type,
                             float wt_ptDist_loss, float wt_inlier_loss, float loss_discount_factor, float* transforms,
                             double* losses, float* grad_logits) {
  return dsir_align_loss_backward2(c, pt_src, pt_ref, idx, logits, labels, transform_gt, pairs, J, K, n_iter, loss_type, wt_ptDist_loss,
                                   wt_inlier_loss, loss_discount_factor, transforms, losses, grad_logits, nullptr);
}

int dsir_graph_stats(dsir_ctx* c, int64_t* out) {
  if (!c || !out) return 1;
  for (int i = 0; i < 4; ++i) out[i] = c->graph_nodes[i];
  return 0;
}

int dsir_enable_graph(dsir_ctx* c, int enable) {
  if (!c) return 1;
  c->use_graph = enable != 0;
  if (!c->use_graph) c->drop_graphs();
  return 0;
}

int dsir_voxel_downsample(dsir_ctx* c, const float* points, const int64_t* offsets, int clouds, int stride, float voxel_size,
                          const float* crop, int cap, float* out, int32_t* counts) {
  if (!c) return 1;
  if (!points || !offsets || !out || !counts || clouds < 1 || clouds > 1000 || stride < 3 || stride > 16 || cap < 1 ||
      !(voxel_size > 0.f))
    return fail(c, "dsir_voxel_downsample: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  const int64_t total = offsets[clouds];
  if (total <= 0 || total > 0x7fffffffll) return fail(c, "dsir_voxel_downsample: %lld points unsupported", (long long)total);
  c->ws.top = 0; c->ws.overflow = false;
  void* scratch = c->ws.raw(voxel_downsample_scratch_bytes(total, clouds));
  if (c->ws.overflow) return fail(c, "workspace too small for %lld raw points (raise max_points / max_pairs)", (long long)total);
  // the host offsets are consumed by an async copy: make the call self-contained
  HIP_OK(c, hipStreamSynchronize(c->stream));
  if (int r = launch_voxel_downsample(points, offsets, clouds, stride, voxel_size, crop, cap, out, counts, scratch, c->stream))
    return fail(c, "dsir_voxel_downsample: launch failed (%d)", r);
  HIP_OK(c, hipStreamSynchronize(c->stream));
  return post(c);
}

int dsir_resample(dsir_ctx* c, const float* in, const int32_t* counts, int clouds, int cap, int stride, int k, int mode,
                  uint64_t seed, float* out) {
  if (!c) return 1;
  if (!in || !counts || !out || clouds < 1 || cap < 1 || stride < 1 || k < 1 || (mode != 0 && mode != 1))
    return fail(c, "dsir_resample: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  c->ws.top = 0; c->ws.overflow = false;
  void* scratch = c->ws.raw(resample_scratch_bytes(clouds, cap));
  if (c->ws.overflow) return fail(c, "workspace too small for dsir_resample");
  if (int r = launch_resample(in, counts, clouds, cap, stride, k, mode, seed, out, scratch, c->stream))
    return fail(c, "dsir_resample: launch failed (%d)", r);
  return post(c);
}

int dsir_eval_metrics(dsir_ctx* c, const float* pred_T, int64_t pred_stride, const float* gt_T, const float* points_src,
                      const float* points_ref, int pairs, int n, int stride, float rte_thresh, float rre_thresh,
                      double* out) {
  if (!c) return 1;
  if (!pred_T || !gt_T || !points_src || !points_ref || !out || pairs < 1 || n < 1 || stride < 3 || pred_stride < 12)
    return fail(c, "dsir_eval_metrics: bad arguments");
  HIP_OK(c, hipSetDevice(c->device));
  launch_eval_metrics(pred_T, pred_stride, gt_T, points_src, points_ref, pairs, n, stride, rte_thresh, rre_thresh, out,
                      c->stream);
  return post(c);
}

static int match_ts_reset(dsir_ctx* c) {
  std::vector<unsigned long long> init(2 * kMatchSlots);
  for (size_t i = 0; i < kMatchSlots; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
  HIP_OK(c, hipMemcpy(c->match_ts, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  c->match_ts_used = 0;
  return 0;
}
int dsir_enable_match_timer(dsir_ctx* c, int enable) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  c->time_match = enable != 0;
  if (c->time_match) {
    if (!c->match_ts) HIP_OK(c, hipMalloc((void**)&c->match_ts, 2 * kMatchSlots * sizeof(unsigned long long)));
    if (int r = match_ts_reset(c)) return r;
  }
  return 0;
}
int dsir_match_timer_device(dsir_ctx* c, int reset, double* total_ms, int64_t* launches) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  if (c->match_ts && c->match_ts_used) {
    std::vector<unsigned long long> h(2 * c->match_ts_used);
    HIP_OK(c, hipMemcpy(h.data(), c->match_ts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    for (size_t i = 0; i < c->match_ts_used; ++i)
      if (h[2 * i] != ~0ull && h[2 * i + 1] > h[2 * i]) {
        c->match_dev_ms += (double)(h[2 * i + 1] - h[2 * i]) / (double)khz;
        ++c->match_dev_launches;
      }
    if (int r = match_ts_reset(c)) return r;
  }
  if (total_ms) *total_ms = c->match_dev_ms;
  if (launches) *launches = c->match_dev_launches;
  if (reset) { c->match_dev_ms = 0.0; c->match_dev_launches = 0; }
  return 0;
}

int dsir_prune_stats(dsir_ctx* c, int reset, int64_t* out) {
  if (!c || !out) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  unsigned long long h[2];
  HIP_OK(c, hipMemcpy(h, c->screen_acc + 4, sizeof h, hipMemcpyDeviceToHost));
  out[0] = (int64_t)h[0]; out[1] = (int64_t)h[1];
  if (reset) HIP_OK(c, hipMemset(c->screen_acc + 4, 0, sizeof h));
  return 0;
}

int dsir_screen_stats(dsir_ctx* c, int reset, int64_t* out) {
  if (!c || !out) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  unsigned long long h[4];
  HIP_OK(c, hipMemcpy(h, c->screen_acc, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)h[i];
  out[4] = c->exhaustive_searches;
  if (reset) {
    HIP_OK(c, hipMemset(c->screen_acc, 0, sizeof h));
    c->exhaustive_searches = 0;
  }
  return 0;
}

static int collect_match_events(dsir_ctx* c) {
  HIP_OK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->match_events_used; ++i) {
    float ms = 0.f, kms = 0.f;
    const auto& e = c->match_events[i];
    if (hipEventElapsedTime(&ms, e.op0, e.op1) == hipSuccess && hipEventElapsedTime(&kms, e.k0, e.k1) == hipSuccess) {
      c->match_ms += ms; c->match_kernel_ms += kms; ++c->match_launches;
    }
  }
  c->match_events_used = 0;
  return 0;
}

int dsir_match_timer(dsir_ctx* c, int reset, double* total_ms, int64_t* launches) {
  if (!c) return 1;
  if (int r = collect_match_events(c)) return r;
  if (total_ms) *total_ms = c->match_ms;
  if (launches) *launches = c->match_launches;
  if (reset) { c->match_ms = 0.0; c->match_kernel_ms = 0.0; c->match_launches = 0; }
  return 0;
}

int dsir_match_timer2(dsir_ctx* c, int reset, double* op_ms, double* kernel_ms, int64_t* launches) {
  if (!c) return 1;
  if (int r = collect_match_events(c)) return r;
  if (op_ms) *op_ms = c->match_ms;
  if (kernel_ms) *kernel_ms = c->match_kernel_ms;
  if (launches) *launches = c->match_launches;
  if (reset) { c->match_ms = 0.0; c->match_kernel_ms = 0.0; c->match_launches = 0; }
  return 0;
}

void dsir_split_f16(const float* x, int64_t n, uint16_t* hi, uint16_t* lo) {
  if (x && hi && lo && n > 0) split_weights_f16(x, (size_t)n, hi, lo);
}

int dsir_enable_agg_split(dsir_ctx* c, int enable) {
  if (!c) return 1;
  c->agg_split = enable != 0;
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

int dsir_set_prune_thresholds(dsir_ctx* c, int min_points, int64_t min_rows) {
  if (!c) return 1;
  c->prune_min_points = min_points > 0 ? min_points : 0;
  c->prune_min_rows = min_rows > 0 ? min_rows : 0;
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

int dsir_set_kabsch_chunked_min(dsir_ctx* c, int min_points) {
  if (!c) return 1;
  c->kabsch_chunked_min = min_points > 0 ? min_points : 0;
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

int dsir_walk_trace(dsir_ctx* c, int reset, int64_t* out, int64_t* clock_khz) {
  if (!c) return 1;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)dsir_ctx::kWalkSlots * kWalkMaxPhases * 4;
  if (!c->walk_trace) return fail(c, "dsir_walk_trace: tracing is off (DSIR_TUNING=1 DSIR_WALK_TRACE=1 before dsir_create)");
  if (out) HIP_OK(c, hipMemcpy(out, c->walk_trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) {
    std::vector<unsigned long long> init(n);
    for (size_t i = 0; i < n; ++i) init[i] = (i & 3) < 2 ? ~0ull : 0ull;      // two minima, two maxima
    HIP_OK(c, hipMemcpy(c->walk_trace, init.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  if (clock_khz) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    *clock_khz = khz;
  }
  return 0;
}

int dsir_enable_fork(dsir_ctx* c, int enable) {
  if (!c) return 1;
  c->fork_mode = enable != 0;
  if (c->fork_mode) ensure_aux_streams(c);
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

int dsir_enable_walk(dsir_ctx* c, int enable) {
  if (!c) return 1;
  c->walk_mode = enable != 0;
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

int dsir_enable_screen(dsir_ctx* c, int enable) {
  if (!c) return 1;
  c->screen_mode = enable != 0;
  // a captured registration has the choice baked in
  c->drop_graphs();
  return 0;
}

}  // extern "C"
