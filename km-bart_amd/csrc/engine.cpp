// Host-side driver of the KM-BART hot path: owns the parameter census, lays activations out in the
// caller's workspace and enqueues the HIP kernels of one training step / decode step on a stream.
// No device allocation, no synchronisation: the Python host (torch) owns memory and streams.
//
// Reference call path being replaced:
//   src/training.py:118-143 -> src/model/model.py:325-405 -> src/model/modules.py:104-165
//   + transformers 3.0.2 EncoderLayer / BartDecoder / AdamW.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>
#include "kernels.h"
#include "diag.h"

namespace {

thread_local std::string g_err;
int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define KCHK(expr)                 \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != 0) return rc_;      \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// fp32 validation mode (kmb_set_precision): activations are float, the GEMM / attention / LayerNorm / embedding launches
// go to the plain fp32 kernels of fp32_validate.hip.  Activation pointers keep their bf16_t* type in the host code;
// EP() advances them by ELEMENTS of the current width.  Set for the duration of one forward call (a handle is
// single-threaded, include/kmbart.h).
thread_local bool g_f32 = false;
// split-K slab of the caller's stream for small-batch forward / data-gradient GEMMs (set for the duration of one call)
thread_local float* g_small_slab = nullptr;
thread_local size_t g_small_floats = 0;
inline size_t esz() { return g_f32 ? 4 : 2; }
// KMB_FP32_HEAD=1: fp32 logits + the register-resident fp32 cross-entropy in the bf16 product mode as well.  Read ONCE and
// used by the workspace layout and by the forward pass alike (the logits buffer is sized for whichever head runs).
inline bool force_fp32_head() {
  static const bool on = getenv("KMB_FP32_HEAD") != nullptr && getenv("KMB_FP32_HEAD")[0] == '1';
  return on;
}
template <typename T> inline T* EP(T* p, size_t n) { return (T*)((char*)p + n * esz()); }

struct ParamInfo { std::string name; size_t off; int rows, cols; };
struct AttnP { size_t qkv_w, qkv_b, o_w, o_b, ln_g, ln_b; };
// ca: the cross-attention block; its qkv_w / qkv_b are the QUERY projection only -- the key | value projections of all
// decoder layers live together in the arena (kmb_handle::xkv_w / xkv_b), ca_kv_w / ca_kv_b are this layer's slices
struct LayerP { AttnP sa, ca; size_t ca_kv_w, ca_kv_b, fc1_w, fc1_b, fc2_w, fc2_b, ln_g, ln_b; };
struct Bucket { size_t off, count; };
struct HeadP { size_t dw = 0, db = 0, ow = 0, ob = 0; int d_in = 0, C = 0; bool on = false; };

struct EncAct { bf16_t *qkv, *o, *z1, *y1, *u, *hh, *z2; float *lse, *m1, *r1, *m2, *r2; };
struct DecAct {
  bf16_t *qkv, *o1, *z1, *y1, *cq, *ckv, *o2, *z2, *y2, *u, *hh, *z3;
  float *lse1, *lse2, *m1, *r1, *m2, *r2, *m3, *r3;
};

class Bump {
 public:
  Bump(char* base, size_t cap) : base_(base), cap_(cap), off_(0) {}
  template <typename T> T* take(size_t n) {
    off_ = align_up(off_, 256);
    T* p = reinterpret_cast<T*>(base_ + off_);
    off_ += n * sizeof(T);
    return p;
  }
  // n activation elements of the current precision (bf16, or float in the fp32 validation mode)
  bf16_t* act(size_t n) { return reinterpret_cast<bf16_t*>(take<char>(n * esz())); }
  size_t used() const { return align_up(off_, 256); }
  bool ok() const { return base_ == nullptr || off_ <= cap_; }
 private:
  char* base_; size_t cap_, off_;
};

}  // namespace

struct kmb_handle {
  kmb_config cfg;
  int d, He, Hd, Fe, Fd, V, Vpad, Fin, Fpad, Prows;
  std::vector<ParamInfo> params;
  size_t arena = 0;
  size_t img_w, img_b, enc_pos, enc_lne_g, enc_lne_b, dec_pos, dec_lne_g, dec_lne_b, shared;
  // Cross-attention keys and values are projections of the ENCODER output: the same input for every decoder layer, known
  // before the decoder starts.  Their weights [Ld][k | v][d, d] and biases [Ld][k | v][d] are contiguous in the arena so
  // that ONE GEMM computes all layers' keys | values ([Me, Ld * 2d]), ONE data-gradient GEMM (K = Ld * 2d) produces the
  // encoder-output gradient (instead of six launches that each re-read and re-write it) and ONE weight-gradient GEMM
  // their gradients.  Parameter names are the reference's (model.decoder.layers.N.encoder_attn.k_proj.weight ...).
  size_t xkv_w = 0, xkv_b = 0;
  std::vector<LayerP> enc, dec;
  HeadP head[3];                    // mrm, attribute, relation (src/model/model.py:133-158)
  size_t heads_begin = 0, heads_end = 0; int head_rows_cap = 0;
  std::vector<Bucket> buckets;      // in backward completion order
  bool enc_given = false;           // the last forward started from the caller's encoder states (kmb_forward_opts)
  std::vector<hipEvent_t> events;
  // bound memory
  float *P = nullptr, *G = nullptr, *M1 = nullptr, *M2 = nullptr, *flb = nullptr;
  bf16_t* PB = nullptr;
  bf16_t* imgw_pad = nullptr;       // inside the bf16 arena tail: [d, Fpad]
  char* ws = nullptr; size_t ws_bytes = 0;
  uint64_t seed = 0x5eedULL; uint64_t step = 0;
  bool fp32 = false;    // kmb_set_precision(1): fp32 validation forward
  bool have_hdec = false;   // xd[Ld] of the last forward is still in the workspace (kmb_last_logits)
  int lm_chunk = 8192;  // rows of fp32 logits per LM-head launch (bounds the logits buffer at 1.65 GB for V = 50320)
  // ---- state of the last forward (consumed by backward)
  kmb_batch bt{}; bool have_fwd = false; bool fwd_train = false; bool have_bwd = false;
  int Me = 0, Md = 0, Ntot = 0;
  std::vector<EncAct> ea; std::vector<DecAct> da;
  std::vector<bf16_t*> xe, xd;
  bf16_t *ze0 = nullptr, *zd0 = nullptr; float *me0, *re0, *md0, *rd0;
  bf16_t* xf = nullptr; float* img_emb = nullptr; int32_t* img_src = nullptr; bf16_t* dimg = nullptr;
  float* logits_c = nullptr; size_t logits_c_floats = 0; bf16_t* dlogits_c = nullptr; float* loss_rows = nullptr; int32_t* count = nullptr;
  int32_t* status = nullptr; float* loss_dev = nullptr;
  bf16_t *dhdec, *dyA, *dyB, *dz, *dob, *denc;
  bf16_t *ckv_all = nullptr, *dckv_all = nullptr;   // [Me, Ld * 2d]: every decoder layer's cross-attention k | v and their gradients
  // tied-head cross-entropy without a pass over the logits (loss.hip): per-row shift / picked label value / row sum / scale,
  // the row-scaled decoder states a . H and the bias padded to Vpad
  float *ce_shift = nullptr, *ce_pick = nullptr, *ce_srow = nullptr, *ce_alpha = nullptr, *ce_bias = nullptr;
  bf16_t* ce_ah = nullptr;
  // gradient buffers read by the weight-gradient GEMMs of the side stream: one per LayerNorm site
  // (0 = FFN, 1 = self-attention, 2 = cross-attention) and per layer parity, so that the main stream can run
  // up to one layer ahead of the side stream without overwriting what it still reads
  // parts[site]: partial sums of the parameter gradients a site reduces (0 FFN LayerNorm, 1 fc1 bias column sums, 2 self-attn
  // LayerNorm, 3 self-attn q|k|v bias, 4 cross-attn LayerNorm, 5 cross-attn q|k|v bias): their reducers run on the side
  // stream too, so the partials need the same lifetime as the gradient buffers above
  struct BwdBufs { bf16_t *dz[3], *dsub[3], *du, *dqkv, *dcq, *dckv; float* parts[6]; } bb[2];
  hipStream_t side = nullptr; bool side_on = true;
  // grouped weight gradients (wgrad_side / wgrad_flush): a layer's problems wait here until the layer's last one is known
  std::vector<KmbGemm> wg_pending; bool wg_group = false;
  std::vector<hipEvent_t> ring; size_t ring_pos = 0;
  std::vector<hipEvent_t> layer_done;   // recorded on the side stream
  hipEvent_t head_wgrad_done = nullptr; bool head_wgrad_pending = false;
  // ---- native data parallelism (kmb_comm_*)
  ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 0;
  hipStream_t comm_stream = nullptr; hipEvent_t comm_ev = nullptr;
  int64_t comm_piece_cap = 0;   // piece size of the last algo-1 exchange: the moments' shards follow its piece boundaries
  uint64_t mirror_version = 1;   // bumped whenever the bf16 mirror is rewritten (sync / optimizer): kmb_gen_begin repacks the decoder weights only then
  bool moments_sharded = false; // set by an algo-1 exchange with a fused optimizer on more than one rank, cleared by kmb_comm_gather_moments
  float* parts = nullptr;
  // pre-training head scratch
  bf16_t *hx = nullptr, *hy = nullptr, *hdy = nullptr, *hdx = nullptr, *hdlg = nullptr; float *hlg = nullptr, *hloss = nullptr, *dhead = nullptr;
  float* losses5 = nullptr;
  float* slab = nullptr; size_t slab_floats = 0;            // split-K partials of the side stream's weight gradients
  float* head_slab = nullptr; size_t head_slab_floats = 0;  // ... of the pre-training heads' (caller's stream)
  float* small_slab = nullptr; size_t small_floats = 0;
  hipEvent_t next_event() { hipEvent_t e = ring[ring_pos]; ring_pos = (ring_pos + 1) % ring.size(); return e; }
  // ---- generation state
  struct Gen {
    bool active = false; int B = 0, S = 0, nb = 0, R = 0, Tmax = 0;
    kmb_batch bt{};
    std::vector<bf16_t*> ckv;              // per layer [B*S, 2d]
    std::vector<bf16_t*> kc[2], vc[2];     // per layer self caches, double buffered [R, Tmax, d]
    int cur = 0;
    int32_t *kv_row = nullptr;             // [R] -> batch item (the current one of the two copies at kv_row_base)
    int32_t *kv_row_base = nullptr;
    bf16_t *x0, *x1, *qkv, *o, *z, *y, *cq, *u, *hh; float *mean, *rstd; float* slab;
    std::vector<bf16_t*> wp;               // fused decode blocks: fragment-order weight copies, 6 per layer (empty: not eligible)
    // the last kmb_gen_step's final decoder states: normalised rows at last_x, or (fused blocks, no vocabulary projection)
    // pre-LayerNorm sums at last_z with the last layer's LayerNorm (last_g, last_b) still to be applied
    const bf16_t* last_x = nullptr; const bf16_t* last_z = nullptr; const float *last_g = nullptr, *last_b = nullptr;
    uint32_t* bars = nullptr;
    // History index of the self-attention caches (round 5): position t of beam row r lives in cache row hist[r][t].  A beam reorder
    // (mixins.py:419-434 _reorder_cache) permutes these [R, Tmax] int rows into the other copy instead of gathering every layer's
    // K and V cache ([R, t, d] x 12 buffers: 5-40 us per decode step at batch 64 x 5 beams); the caches are never copied
    // (kc[0] / vc[0] only).  KMB_GEN_HIST=0 restores the physical reorder.
    int32_t* hist[2] = {nullptr, nullptr}; int hcur = 0; bool use_hist = true;
    // per-block (maximum, sum-exp) pairs the last kmb_gen_step's vocabulary projection left beside the logits at head_stats_for
    // (head_stats_blocks column blocks; 0: none -- the step ran another GEMM kernel, or no projection): kmb_gen_beam_step selects from them
    float* head_stats = nullptr; int head_stats_blocks = 0; const float* head_stats_for = nullptr;
    // x0 already holds the embedded rows of decode step x0_step for the tokens at x0_tokens (kmb_gen_beam_step embedded the tokens it chose
    // in its own launch): the kmb_gen_step of exactly that step and token buffer skips its embedding launch.  -1: no
    int x0_step = -1; const int64_t* x0_tokens = nullptr;
    uint64_t packed_version = 0; const bf16_t* packed_at = nullptr;   // the fragment-order copies at wp[0] were made from mirror version ...
  } gen;

  KmbDrop drop_site(int site, bool train) const {
    KmbDrop dr{0u, 0u, 1.f};
    if (!train || cfg.dropout <= 0.f) return dr;
    uint32_t thr = (uint32_t)lrintf(cfg.dropout * 65536.f);
    if (thr > 65535u) thr = 65535u;
    dr.thr16 = thr;
    dr.seed = (uint32_t)splitmix(seed ^ splitmix(step * 0x10001ull + (uint64_t)site));
    dr.scale = 1.f / (1.f - (float)thr / 65536.f);
    return dr;
  }
  bf16_t* wb(size_t off) const { return g_f32 ? reinterpret_cast<bf16_t*>(P + off) : PB + off; }   // GEMM B operand: bf16 mirror (fp32 master in validation mode)
  float* pf(size_t off) const { return P + off; }
  float* gf(size_t off) const { return G + off; }
};

namespace {

struct PrecisionScope {   // g_f32 follows the handle for the duration of one call
  explicit PrecisionScope(const kmb_handle* h) { g_f32 = h->fp32; }
  ~PrecisionScope() { g_f32 = false; g_small_slab = nullptr; g_small_floats = 0; }
};

size_t add_param(kmb_handle* h, const std::string& name, int rows, int cols) {
  h->arena = align_up(h->arena, 64);
  const size_t off = h->arena;
  h->params.push_back({name, off, rows, cols});
  h->arena += (size_t)rows * cols;
  return off;
}

void add_attn(kmb_handle* h, const std::string& p, const std::string& ln, AttnP& a) {
  const int d = h->d;
  a.qkv_w = add_param(h, p + "q_proj.weight", d, d);
  add_param(h, p + "k_proj.weight", d, d);
  add_param(h, p + "v_proj.weight", d, d);
  a.qkv_b = add_param(h, p + "q_proj.bias", 1, d);
  add_param(h, p + "k_proj.bias", 1, d);
  add_param(h, p + "v_proj.bias", 1, d);
  a.o_w = add_param(h, p + "out_proj.weight", d, d);
  a.o_b = add_param(h, p + "out_proj.bias", 1, d);
  a.ln_g = add_param(h, ln + ".weight", 1, d);
  a.ln_b = add_param(h, ln + ".bias", 1, d);
}

// cross-attention block of a decoder layer: query projection, output projection, LayerNorm (k | v: see kmb_handle::xkv_w)
void add_cross_attn(kmb_handle* h, const std::string& p, const std::string& ln, AttnP& a) {
  const int d = h->d;
  a.qkv_w = add_param(h, p + "q_proj.weight", d, d);
  a.qkv_b = add_param(h, p + "q_proj.bias", 1, d);
  a.o_w = add_param(h, p + "out_proj.weight", d, d);
  a.o_b = add_param(h, p + "out_proj.bias", 1, d);
  a.ln_g = add_param(h, ln + ".weight", 1, d);
  a.ln_b = add_param(h, ln + ".bias", 1, d);
}

void add_ffn(kmb_handle* h, const std::string& p, int F, LayerP& L) {
  const int d = h->d;
  L.fc1_w = add_param(h, p + "fc1.weight", F, d);
  L.fc1_b = add_param(h, p + "fc1.bias", 1, F);
  L.fc2_w = add_param(h, p + "fc2.weight", d, F);
  L.fc2_b = add_param(h, p + "fc2.bias", 1, d);
  L.ln_g = add_param(h, p + "final_layer_norm.weight", 1, d);
  L.ln_b = add_param(h, p + "final_layer_norm.bias", 1, d);
}

// ---- optional per-launch timing of the GEMM kernels with HIP events (bench.py roofline leg) ----
struct GemmProfiler {
  bool on = false;
  std::vector<hipEvent_t> ev;        // pairs
  // pair: the event pair that brackets the launch; share: this record's part of that launch's time (1 for a launch of its own; a
  // grouped weight-gradient launch has one record per problem, time split by FLOPs, `launch` 1 on the first of them only)
  struct Rec { int variant; double flops; int M, N, K, split, act, res; size_t pair = 0; float share = 1.f; int launch = 1; int group_n = 1; };
  std::vector<Rec> recs;
  size_t used = 0;
} g_prof;

KmbGemm gemm0() { KmbGemm g; memset(&g, 0, sizeof(g)); g.col_scale = 1.f; g.drop_scale = 1.f; return g; }

int run_gemm(const KmbGemm& g, hipStream_t s) {
  if (g_f32) {
    const char* why32 = kmb_f32_gemm_check(g);
    if (why32) return fail("fp32 validation GEMM: %s", why32);
    HIPCHK(kmb_f32_gemm_launch(g, s));
    return 0;
  }
  const char* why = kmb_gemm_check(g);
  if (why) return fail("%s (M=%d N=%d K=%d lda=%d ldb=%d akc=%d bkc=%d)", why, g.M, g.N, g.K, g.lda, g.ldb, g.a_kc, g.b_kc);
  // Small-batch regime (the reference's default per-GPU batch is 64: M = 2048-4096 rows): a forward / data-gradient
  // GEMM with N = 768 has 96-192 output tiles for 256 CUs and a serial K loop of up to 48 steps.  Split K over
  // workgroups and let one pass sum the slabs and apply the linear layer's epilogue (bias, q-scale, dropout, residual).
  // 2048x768x3072: 51 -> ~20 us stand-alone; inside a step the idle CUs were already running the side stream's weight
  // gradients, so the whole step gains 2 % at b = 64 (8.06 -> 7.87 ms).  Large batches never take this path.
  static const bool small_ok = !(getenv("KMB_SMALL_SPLIT") && getenv("KMB_SMALL_SPLIT")[0] == '0');
  if (small_ok && g_small_slab != nullptr && g.a_kc == 1 && g.split_k <= 1 && g.act == 0 && !g.preact && !g.colsum && !g.aux &&
      !g.out_f32 && g.out_bf16 && g.beta == 0.f && (g.N & 7) == 0 && (g.K % 64) == 0 && g.M > 512) {
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int nt = g.K / 64;
    static const int small_fill = KMB_DIAG_ENV("KMB_SMALL_FILL") ? atoi(KMB_DIAG_ENV("KMB_SMALL_FILL")) : 512;   // A/B knob
    int S = tiles > 0 ? small_fill / tiles : 1;
    // Round 6: only for <= 48 tiles (M <= 1024 rows of N = 768: b <= 16 encoder rows, b <= 32 decoder rows).  Above that the four-stage 128 x 128
    // kernel (variant 5) -- whose K loop lost its drained prefetch and its accumulator shuffles this round -- runs a lone workgroup's whole K loop
    // faster than five slices + the reduction pass: same box, alternating processes, two rounds (KMB_SMALL_SPLIT=1 | 0): b = 48 6.35-6.37 -> 6.18-6.24 ms,
    // b = 64 6.83-6.91 -> 6.57-6.74, b = 96 8.44-8.52 -> 8.13-8.15; b = 32 and 128 equal; b = 16 4.80-4.86 with the split, 5.05-5.11 without.
    if (tiles > 48) S = 1;
    if (S > 8) S = 8;
    if (S > nt / 4) S = nt / 4;
    while (S > 1 && (size_t)S * g.M * g.N > g_small_floats) --S;
    if (S >= 3) {   // two slices do not pay for the extra pass (b = 128: 10.98 vs 10.92 ms); 96-tile shapes get five
      KmbGemm q = g;
      q.split_k = S; q.slab = g_small_slab; q.out_bf16 = nullptr; q.bias = nullptr; q.residual = nullptr;
      q.drop_thr16 = 0u; q.col_scale = 1.f; q.col_scale_n = 0;
      KCHK(run_gemm(q, s));
      const KmbDrop dr{g.drop_thr16, g.drop_seed, g.drop_scale};
      HIPCHK(kmb_reduce_slabs_epi_launch(g_small_slab, S, (size_t)g.M * g.N, g.bias, g.col_scale, g.col_scale_n, dr, g.residual,
                                         g.ld_res, g.out_bf16, g.ld_out_bf16, g.M, g.N, s));
      return 0;
    }
  }
  if (g_prof.on) {
    if (g_prof.used + 2 > g_prof.ev.size()) {
      const size_t old = g_prof.ev.size();
      g_prof.ev.resize(old + 512);
      for (size_t i = old; i < g_prof.ev.size(); ++i) HIPCHK(hipEventCreate(&g_prof.ev[i]));
    }
    HIPCHK(hipEventRecord(g_prof.ev[g_prof.used], s));
    HIPCHK(kmb_gemm_launch(g, s));
    HIPCHK(hipEventRecord(g_prof.ev[g_prof.used + 1], s));
    g_prof.recs.push_back({g.a_kc * 2 + g.b_kc, 2.0 * g.M * g.N * (double)g.K, g.M, g.N, g.K, g.split_k, g.act, g.residual != nullptr ? 1 : 0,
                           g_prof.used / 2, 1.f, 1, 1});
    g_prof.used += 2;
    return 0;
  }
  HIPCHK(kmb_gemm_launch(g, s));
  return 0;
}

// Y[M,N] = X[M,K] W[N,K]^T + b
KmbGemm lin_fwd(const bf16_t* x, int ldx, const bf16_t* w, const float* b, int M, int N, int K) {
  KmbGemm g = gemm0();
  g.A = x; g.lda = ldx; g.a_kc = 1; g.B = w; g.ldb = K; g.b_kc = 1; g.M = M; g.N = N; g.K = K; g.bias = b;
  return g;
}
// dX[M,K] = dY[M,N] W[N,K]
KmbGemm lin_dgrad(const bf16_t* dy, int lddy, const bf16_t* w, int M, int N, int K) {
  KmbGemm g = gemm0();
  g.A = dy; g.lda = lddy; g.a_kc = 1; g.B = w; g.ldb = K; g.b_kc = 0; g.M = M; g.N = K; g.K = N;
  return g;
}
// dW[N,K] = dY[M,N]^T X[M,K]   (fp32, written straight into the gradient arena)
KmbGemm lin_wgrad(const bf16_t* dy, int lddy, const bf16_t* x, int ldx, float* dW, int M, int N, int K, float beta) {
  KmbGemm g = gemm0();
  g.A = dy; g.lda = lddy; g.a_kc = 0; g.B = x; g.ldb = ldx; g.b_kc = 0; g.M = N; g.N = K; g.K = M;
  g.out_f32 = dW; g.ld_out_f32 = K; g.beta = beta;
  return g;
}

// Weight-gradient GEMMs have few output tiles (768x768 -> 36) and a very long reduction (all tokens):
// split K over workgroups so that the grid fills the chip; partial slabs are summed by one pass.
// `slab` / `slab_floats`: the partial-sum buffer of the STREAM the launch goes to.  The side stream's weight gradients use
// h->slab; the pre-training heads (head_run, caller's stream, concurrent with the tied matrix's gradient on the side
// stream) have their own h->head_slab -- two streams never share one.
int run_wgrad(kmb_handle* h, KmbGemm g, hipStream_t s, float* slab, size_t slab_floats) {
  const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
  const int nt = (g.K + 63) / 64;
  // Slices fill `fill` workgroup slots: 512 (two 128x128 workgroups per CU) when the reduction is long.  With a short
  // reduction (small batches: <= 8192 encoder tokens; the rule goes by the batch, not by the GEMM: a per-GEMM rule measured
  // worse at b = 128) the slab traffic of many slices -- S x the gradient written, then read --
  // costs more than the fuller grid buys, and the caller's stream keeps the other CUs busy anyway: whole step, same box
  // (tools/step_ab_seq.sh): b = 64 7.52 ms with 256 against 7.78 with 512 (384: 7.67, 192: 7.60), b = 128 10.51 with 384
  // against 10.80 (256: 10.95), b = 32 5.99 with 256 against 6.17, b = 256 16.7 with 512 against 18.3 with 256.
  static const int fill_env = KMB_DIAG_ENV("KMB_WGRAD_FILL") ? atoi(KMB_DIAG_ENV("KMB_WGRAD_FILL")) : 0;   // A/B knob
  const int mmax = h->Me > h->Md ? h->Me : h->Md;   // tokens of the longer side: how busy the caller's stream keeps the chip
  // Round 6: 768 slots from 16384 tokens on (was 512).  With the transposing reads as inline asm in every split-K kernel (no drained prefetch) a
  // slice is cheaper than it was when 512 was measured: same box, alternating processes (KMB_WGRAD_FILL, diagnostic library, two rounds each):
  // b = 256 14.17-14.20 ms with 768 against 14.30-14.38 with 512 (1024: 14.28-14.30, 1536: 15.2), b = 512 24.20-24.34 against 24.46-24.59,
  // b = 1024 44.49-44.57 against 44.61-45.01; b = 128 (8192 tokens: stays at 384) 9.40-9.44 with 768 against 9.17-9.26.
  const int fill = fill_env > 0 ? fill_env : mmax <= 4096 ? 256 : mmax <= 8192 ? 384 : 768;
  int S = fill / tiles;   // floor: a partial last round costs more than it fills
                         // (tools/wgrad_split_sweep.py: 36 tiles S14 59 us vs S11 70 us, 72 tiles S7 97 vs S6 104)
  if (S > 16) S = 16;
  // 128-160 tiles and a very long reduction (3072x768 over 32768 tokens): 256x256 tiles with a slice count that fills
  // the chip once beat the 128x128 kernel at S = 3 by 10-12 % (tools/wgrad_split_sweep.py); with these slices the
  // launcher's timing picks the 256x256 kernel.  (Slice counts rounded to the 8 XCDs -- 3 -> 4, 7 -> 8, 14 -> 16, one slice
  // per XCD under the slice-major enumeration -- measured 18 % slower on the weight gradients: 780 -> 638 TFLOP/s.)
  // (wider ranges gain what the longer slab reduction costs -- unless the reduction is very long: 2304x768 over 65536
  // tokens, 108 tiles: 9 slices of the 256x256 kernel 251 us against 4 of the 128x128 kernel 294)
  static const int slots256 = KMB_DIAG_ENV("KMB_WG256_SLOTS") ? atoi(KMB_DIAG_ENV("KMB_WG256_SLOTS")) : 256;   // A/B knob: workgroup slots the 256 x 256 slices fill
  if (tiles >= 96 && tiles <= 160) {
    const int tiles256 = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    const int s256 = slots256 / tiles256;
    if (s256 > S && ((tiles >= 128 && nt >= 512) || nt / s256 >= 100)) S = s256;
  }
  // More 256x256 tiles than CUs and a poorly filled last round (the tied 50320x768 matrix: 591 tiles = 2.31 rounds, 77 %
  // of three): two or three K slices make the rounds come out even (x 3 = 6.93 of 7).  3137 -> 2480 + 155 us of slab
  // reduction at 32768 tokens (tools/wgrad_split_sweep.py's sibling measurement).
  if (S <= 1 && nt >= 256) {
    const int tiles256 = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    if (tiles256 > 256) {
      auto eff = [&](int k) { const double r = (double)tiles256 * k / 256.0; return r / std::ceil(r); };
      int best = 1;
      for (int k = 2; k <= 3; ++k)
        if (eff(k) > eff(best) + 0.02) best = k;
      if (eff(best) >= eff(1) + 0.10) S = best;
    }
  }
  // 64 .. 128 tiles of 256x256 and no slice from the rules above (the batched cross-attention k | v weights: 9216 x 768 =
  // 108 tiles): two or more slices so that the 256x256 kernel covers the chip once
  if (S <= 1 && nt >= 128) {
    const int tiles256 = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    if (tiles256 >= 64 && tiles256 <= 128) S = slots256 / tiles256;
  }
  if (S > nt / 2) S = nt / 2;
  while (S > 1 && (size_t)S * g.M * g.N > slab_floats) --S;
  if (S <= 1 || slab == nullptr || g.ld_out_f32 != g.N || ((size_t)g.M * g.N & 3)) return run_gemm(g, s);
  float* out = g.out_f32;
  const float beta = g.beta;
  g.split_k = S; g.slab = slab; g.out_f32 = nullptr; g.beta = 0.f;
  KCHK(run_gemm(g, s));
  // KMB_SKIP_SLAB_REDUCE=1 (diagnostic build, TIMING ONLY -- the gradients are never written): what the 63 reduction launches of a
  // step cost where they run (b = 256: 0.62 of 15.85 ms, b = 512: 0.6 of 27.0, b = 1024: 0.6 of 49.7, b = 64: nothing).  Folding the
  // reduction into the GEMM -- the slices of a tile meet at a counter and the last one sums the slabs, write-through stores and
  // L2-bypassing loads so that it is correct across XCDs -- was built in round 5, bit-identical, and SLOWER (3072 x 768 x 8192 in
  // 7 slices: 80 -> 139 us; the step +2.5 % at b = 64, +5 % at 256, +4 % at 1024): slabs written through to memory and read back
  // past the L2 cost several times what the pass over L2- / Infinity-Cache-resident slabs costs (profiles/r05_grouped_weight_gradients.md)
  static const bool skip_reduce = KMB_DIAG_ENV("KMB_SKIP_SLAB_REDUCE") != nullptr;
  if (skip_reduce) return 0;
  HIPCHK(kmb_reduce_slabs_launch(slab, S, (size_t)g.M * g.N, out, (size_t)g.M * g.N, beta, s));
  return 0;
}

int ensure_side(kmb_handle* h) {
  if (!h->side_on || h->side != nullptr) return 0;
  const char* env = getenv("KMB_NO_SIDE_STREAM");
  if (env && env[0] == '1') { h->side_on = false; return 0; }
  {
    // The side stream carries the weight gradients, which nothing in backward waits for; the caller's stream carries the
    // critical path (data gradients, LayerNorm / attention backward, reducers).  KMB_SIDE_PRIORITY = low | high | default
    // picks the side stream's queue priority (experiment knob; default: the device's default priority).
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const char* pr = KMB_DIAG_ENV("KMB_SIDE_PRIORITY");
    if (pr && (pr[0] == 'l' || pr[0] == 'h'))
      HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, pr[0] == 'l' ? least : greatest));
    else
      HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  }
  h->ring.resize(1024);   // more than one backward pass records (~90): an event is never re-recorded while an earlier wait on it may be pending
  for (auto& e : h->ring) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  h->layer_done.resize(h->cfg.encoder_layers + h->cfg.decoder_layers + 2);
  for (auto& e : h->layer_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->head_wgrad_done, hipEventDisableTiming));
  return 0;
}

// Weight gradients are off the critical path (nothing in backward reads them): enqueue them on the side stream
// behind an event that marks "everything the main stream has produced so far".  Two different GEMMs in flight are
// out of phase, so one's output-store burst overlaps the other's matrix work and partial waves get filled.
// Grouped weight gradients (csrc/gemm.hip, gemm_group_wgrad_kernel).  With few tokens a weight gradient has 36-144 output
// tiles and a short reduction: alone it needs a 3- to 7-fold split-K plus a slab reduction to cover the chip -- 63 + 91
// launches a step on the side stream, each GEMM behind an event pair.  A layer's four to six weight gradients together are
// 432-504 tiles: ONE launch, whole reductions, no slabs (16 launches a step).  wgrad_side() parks eligible problems in
// h->wg_pending while h->wg_group is set (kmb_backward: token count under the rule there); wgrad_flush() sends them out
// behind a marker of the caller's stream -- at the end of every layer, before that layer's completion events are recorded
// on the side stream.  The operands are per-layer buffers that stay valid until then (the layer_done rule).
// Measured (same box, alternating processes, profiles/r05_grouped_weight_gradients.md): the step gets faster for up to 48
// samples (b = 8: 5.89 -> 5.69 ms, 16: 5.54 -> 5.26, 32: 5.77 -> 5.59, 48: 6.66 -> 6.56) and slower from 64 on (7.25 -> 7.31,
// b = 128: 9.82 -> 10.10): the side stream's busy time falls from 3.1 to 1.8 ms at b = 64, but a launch that occupies every
// workgroup slot of the chip for 85-160 us holds up the caller's stream more than seven-fold split-K launches of 252
// workgroups did -- and without ANY weight gradient that step still takes 6.33 ms (diagnostic KMB_SKIP_WGRAD), without the
// optimizer 6.44: at that size the step is the caller's stream plus the 0.78 ms of AdamW traffic.
int wgrad_flush(kmb_handle* h, hipStream_t sA) {
  if (h->wg_pending.empty()) return 0;
  std::vector<KmbGemm> ps;
  ps.swap(h->wg_pending);
  const bool on_side = h->side_on && h->side != nullptr;
  hipStream_t s = on_side ? h->side : sA;
  if (on_side) {
    hipEvent_t e = h->next_event();
    HIPCHK(hipEventRecord(e, sA));
    HIPCHK(hipStreamWaitEvent(h->side, e, 0));
  }
  if (ps.size() == 1) return run_wgrad(h, ps[0], s, h->slab, h->slab_floats);   // nothing to group with
  if (g_prof.on) {
    if (g_prof.used + 2 > g_prof.ev.size()) {
      const size_t old = g_prof.ev.size();
      g_prof.ev.resize(old + 512);
      for (size_t i = old; i < g_prof.ev.size(); ++i) HIPCHK(hipEventCreate(&g_prof.ev[i]));
    }
    HIPCHK(hipEventRecord(g_prof.ev[g_prof.used], s));
  }
  HIPCHK(kmb_gemm_group_launch(ps.data(), (int)ps.size(), s));
  if (g_prof.on) {
    HIPCHK(hipEventRecord(g_prof.ev[g_prof.used + 1], s));
    double total = 0.0;
    for (const KmbGemm& g : ps) total += 2.0 * g.M * g.N * (double)g.K;
    for (size_t i = 0; i < ps.size(); ++i) {
      const KmbGemm& g = ps[i];
      const double fl = 2.0 * g.M * g.N * (double)g.K;
      g_prof.recs.push_back({0, fl, g.M, g.N, g.K, 0, 0, 0, g_prof.used / 2, (float)(fl / total), i == 0 ? 1 : 0, (int)ps.size()});
    }
    g_prof.used += 2;
  }
  return 0;
}

int wgrad_side(kmb_handle* h, const KmbGemm& g, hipStream_t sA) {
  if (h->wg_group && !g_f32 && kmb_gemm_group_check(&g, 1) == nullptr) {
    h->wg_pending.push_back(g);
    if ((int)h->wg_pending.size() == KMB_GEMM_GROUP_MAX) return wgrad_flush(h, sA);
    return 0;
  }
  KCHK(wgrad_flush(h, sA));   // (keeps the side stream's order: an ineligible problem goes out behind the parked ones)
  // KMB_SKIP_WGRAD=1 (diagnostic build, TIMING ONLY -- the gradients are wrong): no weight-gradient GEMM, no event.  What
  // the step costs without them bounds what any regrouping of the side stream's work can buy (tools/step_ab.sh)
  static const bool skip = KMB_DIAG_ENV("KMB_SKIP_WGRAD") != nullptr;
  if (skip) return 0;
  if (!h->side_on || h->side == nullptr) return run_wgrad(h, g, sA, h->slab, h->slab_floats);
  hipEvent_t e = h->next_event();
  HIPCHK(hipEventRecord(e, sA));
  HIPCHK(hipStreamWaitEvent(h->side, e, 0));
  KCHK(run_wgrad(h, g, h->side, h->slab, h->slab_floats));
  // KMB_SIDE_SERIALIZE=1 (diagnostic): the caller's stream waits for every weight gradient -- the side stream's
  // launches stay where they are, nothing overlaps (run-to-run reproducibility)
  static const bool serialize = KMB_DIAG_ENV("KMB_SIDE_SERIALIZE") != nullptr;
  if (serialize) {
    hipEvent_t e2 = h->next_event();
    HIPCHK(hipEventRecord(e2, h->side));
    HIPCHK(hipStreamWaitEvent(sA, e2, 0));
  }
  return 0;
}

// diagnostic: KMB_BWD_TRACE=1 checksums intermediate buffers of backward on their own stream (no synchronisation); the
// table is printed by kmb_debug_trace_dump.  Finds the first buffer that differs between two passes.
struct TraceRec { const char* name; int layer; };
std::vector<TraceRec> g_trace;
unsigned long long* g_trace_dev = nullptr;
bool g_trace_on = false;
int g_trace_layer = -1;
int trace(const char* name, const void* p, size_t bytes, hipStream_t s) {
  if (!g_trace_on) return 0;
  static const char* only = KMB_DIAG_ENV("KMB_BWD_TRACE_ONLY");   // substring filter: fewer probes disturb the timing less
  if (only && !strstr(name, only)) return 0;
  if (!g_trace_dev) HIPCHK(hipMalloc(&g_trace_dev, 4096 * sizeof(unsigned long long)));
  if (g_trace.size() >= 4096) return 0;
  HIPCHK(kmb_hash_words_launch(p, bytes, g_trace_dev + g_trace.size(), s));
  g_trace.push_back({name, g_trace_layer});
  return 0;
}

// The reducers that fold partial sums into parameter gradients (LayerNorm gamma / beta, biases) produce nothing backward
// waits for: like the weight-gradient GEMMs they go to the side stream, behind an event that marks the producer of the
// partials on the caller's stream.  They are tiny (4-10 us alone) but sat on the critical path between two data-gradient
// GEMMs, where -- sharing the GPU with a weight-gradient GEMM of the side stream -- each took ~57 us (62 per step:
// rocprofv3 kernel stats of the overlapped step, profiles/r02_kernel_stats_b1024.md).  `parts` must not be rewritten
// before the side stream has read it: the per-site, per-layer-parity buffers of BwdBufs.
// Measured (whole step, same box, alternating processes, tools/step_ab_seq.sh): b = 1024 52.9 / 53.1 ms against 53.5 / 53.9
// with the reducers on the caller's stream; b = 256 16.50 / 16.55 against 16.21 / 16.39 -- the other way round (short
// backward: the extra events cost more than the reducers) -- so only long batches take this path.
hipStream_t reducer_stream(kmb_handle* h, hipStream_t sA) {
  static const char* env = KMB_DIAG_ENV("KMB_REDUCERS_ON_MAIN");   // "1": never on the side stream, "0": always (A/B knob)
  const bool want = env ? env[0] == '0' : (h->Me > h->Md ? h->Me : h->Md) >= 16384;
  if (!want || !h->side_on || h->side == nullptr) return sA;
  hipEvent_t e = h->next_event();
  if (hipEventRecord(e, sA) != hipSuccess || hipStreamWaitEvent(h->side, e, 0) != hipSuccess) return sA;
  return h->side;
}

int bias_grad(kmb_handle* h, const bf16_t* dy, int ld, int M, int N, float* out, hipStream_t s) {
  HIPCHK(kmb_colsum_launch(dy, ld, M, N, h->parts, s));
  HIPCHK(kmb_reduce_parts_launch(h->parts, kmb_colsum_parts(M), N, out, N, s));
  return 0;
}

constexpr size_t NO_BIAS = (size_t)-1;
// bias_off: gradient slot of the bias of the linear that produced the (dropped) sub-layer output, or NO_BIAS
// site_parts: this site's own partials buffer (the reducer then runs on the side stream), or nullptr: the shared scratch,
// reducer on the caller's stream
int ln_backward(kmb_handle* h, const bf16_t* dy, const bf16_t* z, const float* mean, const float* rstd, size_t g_off,
                size_t b_off, bf16_t* dz, bf16_t* out2, KmbDrop dy_drop, KmbDrop out2_drop, int M, hipStream_t s,
                size_t bias_off = NO_BIAS, float* site_parts = nullptr) {
  const int d = h->d;
  if (b_off != g_off + (size_t)d) return fail("LayerNorm weight/bias are not adjacent in the arena");
  float* parts = site_parts ? site_parts : h->parts;
  HIPCHK(kmb_ln_bwd_launch(dy, z, mean, rstd, h->pf(g_off), dz, out2, dy_drop, out2_drop, parts, M, d, s));
  const int np = kmb_ln_bwd_parts(M);
  hipStream_t rs = site_parts ? reducer_stream(h, s) : s;
  // partials are [np][3][d]: dgamma | dbeta (adjacent in the arena too: one reduce) | column sums of the sub-layer gradient
  if (bias_off != NO_BIAS) HIPCHK(kmb_reduce_parts2_launch(parts, np, 3 * d, h->gf(g_off), 2 * d, h->gf(bias_off), d, rs));
  else HIPCHK(kmb_reduce_parts_launch(parts, np, 3 * d, h->gf(g_off), 2 * d, rs));
  return 0;
}

// scratch for partial reductions: LayerNorm [<=1024][3][d], attention bias partials [B][3d],
// GEMM column sums [ceil(M/128)][F], colsum kernel [<=64][maxN]
size_t parts_floats(const kmb_handle* h, int Mmax, int B) {
  const size_t d = h->d;
  size_t maxN = 3 * d;
  if ((size_t)h->Fe > maxN) maxN = h->Fe;
  if ((size_t)h->Fd > maxN) maxN = h->Fd;
  size_t need = (size_t)kmb_ln_bwd_parts(Mmax) * 3 * d;
  const size_t attn = (size_t)B * 3 * d;
  const size_t gsum = ((size_t)Mmax + 63) / 64 * maxN;
  const size_t csum = (size_t)64 * maxN;
  if (attn > need) need = attn;
  if (gsum > need) need = gsum;
  if (csum > need) need = csum;
  return need + 1024;
}

// ------------------------------------------------------------------ workspace layout (training)
// With base == nullptr this only measures.
size_t layout_train(kmb_handle* h, char* base, size_t cap, int B, int S, int T, int Ntot, bool assign) {
  const int d = h->d, Fe = h->Fe, Fd = h->Fd;
  const size_t Me = (size_t)B * S, Md = (size_t)B * T;
  const size_t Mmax = Me > Md ? Me : Md;
  const int Le = h->cfg.encoder_layers, Ld = h->cfg.decoder_layers;
  Bump bp(base, cap);
  auto* H = h;
  int32_t* status = bp.take<int32_t>(4);
  int32_t* count = bp.take<int32_t>(4);
  float* loss_dev = bp.take<float>(4);
  bf16_t* xf = bp.act((size_t)(Ntot > 0 ? Ntot : 1) * h->Fpad);
  float* img_emb = bp.take<float>((size_t)(Ntot > 0 ? Ntot : 1) * d);
  bf16_t* dimg = bp.act((size_t)(Ntot > 0 ? Ntot : 1) * d);
  int32_t* img_src = bp.take<int32_t>(Me);
  bf16_t* ze0 = bp.act(Me * d);
  float* me0 = bp.take<float>(Me); float* re0 = bp.take<float>(Me);
  bf16_t* zd0 = bp.act(Md * d);
  float* md0 = bp.take<float>(Md); float* rd0 = bp.take<float>(Md);
  std::vector<bf16_t*> xe(Le + 1), xd(Ld + 1);
  for (int l = 0; l <= Le; ++l) xe[l] = bp.act(Me * d);
  for (int l = 0; l <= Ld; ++l) xd[l] = bp.act(Md * d);
  std::vector<EncAct> ea(Le);
  for (int l = 0; l < Le; ++l) {
    EncAct& a = ea[l];
    a.qkv = bp.act(Me * 3 * d); a.o = bp.act(Me * d); a.z1 = bp.act(Me * d);
    a.y1 = bp.act(Me * d); a.u = bp.act(Me * Fe); a.hh = bp.act(Me * Fe);
    a.z2 = bp.act(Me * d);
    a.lse = bp.take<float>((size_t)B * h->He * S);
    a.m1 = bp.take<float>(Me); a.r1 = bp.take<float>(Me); a.m2 = bp.take<float>(Me); a.r2 = bp.take<float>(Me);
  }
  std::vector<DecAct> da(Ld);
  bf16_t* ckv_all = bp.act(Me * (size_t)Ld * 2 * d);    // [Me, Ld * 2d]: layer l's k | v are columns [l * 2d, (l + 1) * 2d)
  bf16_t* dckv_all = bp.act(Me * (size_t)Ld * 2 * d);
  for (int l = 0; l < Ld; ++l) {
    DecAct& a = da[l];
    a.qkv = bp.act(Md * 3 * d); a.o1 = bp.act(Md * d); a.z1 = bp.act(Md * d);
    a.y1 = bp.act(Md * d); a.cq = bp.act(Md * d); a.ckv = EP(ckv_all, (size_t)l * 2 * d);
    a.o2 = bp.act(Md * d); a.z2 = bp.act(Md * d); a.y2 = bp.act(Md * d);
    a.u = bp.act(Md * Fd); a.hh = bp.act(Md * Fd); a.z3 = bp.act(Md * d);
    a.lse1 = bp.take<float>((size_t)B * h->Hd * T); a.lse2 = bp.take<float>((size_t)B * h->Hd * T);
    a.m1 = bp.take<float>(Md); a.r1 = bp.take<float>(Md); a.m2 = bp.take<float>(Md); a.r2 = bp.take<float>(Md);
    a.m3 = bp.take<float>(Md); a.r3 = bp.take<float>(Md);
  }
  // bf16 product path: the training logits live in dlogits_c (bf16, turned into their own gradient in place by the CE
  // kernel); logits_c only holds the split-K slabs of the head's data gradient (<= 8 x Md x d floats).  The fp32
  // validation mode keeps fp32 logits here, in row chunks of lm_chunk.
  const size_t CH = Md < (size_t)h->lm_chunk ? Md : (size_t)h->lm_chunk;
  const size_t lc_floats = (g_f32 || h->Vpad > 65536 || force_fp32_head()) ? std::max(CH * h->Vpad, (size_t)8 * Md * d) : (size_t)8 * Md * d;
  float* logits_c = bp.take<float>(lc_floats);
  bf16_t* dlogits_c = bp.act(Md * h->Vpad);   // all rows: the head's dgrad / wgrad run once, un-chunked
  // split-K partial slabs of the weight-gradient GEMMs: 14 slices of a 768x768 matrix ... 3 of the tied V x d matrix
  // (the latter only when the batch is long enough for run_wgrad to split it: >= 256 K steps = 16384 decoder tokens)
  const size_t slab_floats = std::max((size_t)20 << 20, Md >= 16384 ? (size_t)3 * h->V * d : (size_t)0);
  float* slab = bp.take<float>(slab_floats);
  // slabs of the small-batch split-K forward / dgrad GEMMs (run_gemm) on the caller's stream; only small batches use them
  const size_t small_floats = Mmax <= 8192 ? (size_t)8 * Mmax * d : 1024;
  float* small_slab = bp.take<float>(small_floats);
  float* loss_rows = bp.take<float>(Md);
  float* ce_shift = bp.take<float>(Md); float* ce_pick = bp.take<float>(Md);
  float* ce_srow = bp.take<float>(Md); float* ce_alpha = bp.take<float>(Md);
  float* ce_bias = bp.take<float>(h->Vpad);
  bf16_t* ce_ah = bp.act(Md * d);
  bf16_t* dhdec = bp.act(Md * d);
  bf16_t* dyA = bp.act(Mmax * d); bf16_t* dyB = bp.act(Mmax * d);
  bf16_t* dz = bp.act(Mmax * d);
  const int Fmax = Fe > Fd ? Fe : Fd;
  kmb_handle::BwdBufs bb[2];
  for (int k = 0; k < 2; ++k) {
    for (int site = 0; site < 3; ++site) {
      bb[k].dz[site] = bp.act(Mmax * d);
      bb[k].dsub[site] = bp.act(Mmax * d);
    }
    bb[k].du = bp.act(Mmax * Fmax);
    bb[k].dqkv = bp.act(Mmax * 3 * d);
    bb[k].dcq = bp.act(Md * d);
    bb[k].dckv = nullptr;   // (the k | v gradients of every layer go to dckv_all)
    for (int site = 0; site < 6; ++site) bb[k].parts[site] = bp.take<float>(parts_floats(h, (int)Mmax, B));
  }
  bf16_t* dob = bp.act(Mmax * d); bf16_t* denc = bp.act(Me * d);
  float* parts = bp.take<float>(parts_floats(h, (int)Mmax, B));
  // pre-training head scratch (only when heads exist and rows were reserved)
  bf16_t *hx = nullptr, *hy = nullptr, *hdy = nullptr, *hdx = nullptr, *hdlg = nullptr;
  float *hlg = nullptr, *hloss = nullptr, *dhead = nullptr, *head_slab = nullptr;
  size_t head_slab_floats = 0;
  float* losses5 = bp.take<float>(8);
  if ((h->head[0].on || h->head[1].on || h->head[2].on) && h->head_rows_cap > 0) {
    const size_t n = (size_t)h->head_rows_cap;
    size_t Cpad = 8;
    for (int k = 0; k < 3; ++k)
      if (h->head[k].on && align_up((size_t)h->head[k].C, 8) > Cpad) Cpad = align_up((size_t)h->head[k].C, 8);
    hx = bp.act(n * 2 * d); hy = bp.act(n * d); hdy = bp.act(n * d);
    hdx = bp.act(n * 2 * d); hdlg = bp.act(n * Cpad); hlg = bp.take<float>(n * Cpad);
    hloss = bp.take<float>(n); dhead = bp.take<float>(Md * d);
    // run_wgrad picks S <= 512 / tiles128 slices: S x M x N stays under 512 x 128 x 128 floats (+ edge-tile slack)
    head_slab_floats = (size_t)9 << 20;
    head_slab = bp.take<float>(head_slab_floats);
  }
  if (assign) {
    H->status = status; H->count = count; H->loss_dev = loss_dev; H->xf = xf; H->img_emb = img_emb; H->dimg = dimg;
    H->img_src = img_src; H->ze0 = ze0; H->me0 = me0; H->re0 = re0; H->zd0 = zd0; H->md0 = md0; H->rd0 = rd0;
    H->xe = xe; H->xd = xd; H->ea = ea; H->da = da; H->logits_c = logits_c; H->logits_c_floats = lc_floats; H->dlogits_c = dlogits_c;
    H->loss_rows = loss_rows; H->dhdec = dhdec; H->dyA = dyA; H->dyB = dyB; H->dz = dz;
    H->bb[0] = bb[0]; H->bb[1] = bb[1]; H->dob = dob; H->denc = denc; H->parts = parts;
    H->slab = slab; H->slab_floats = slab_floats; H->small_slab = small_slab; H->small_floats = small_floats;
    H->hx = hx; H->hy = hy; H->hdy = hdy; H->hdx = hdx; H->hdlg = hdlg; H->hlg = hlg; H->hloss = hloss; H->dhead = dhead;
    H->losses5 = losses5; H->head_slab = head_slab; H->head_slab_floats = head_slab_floats;
    H->ckv_all = ckv_all; H->dckv_all = dckv_all;
    H->ce_shift = ce_shift; H->ce_pick = ce_pick; H->ce_srow = ce_srow; H->ce_alpha = ce_alpha; H->ce_bias = ce_bias; H->ce_ah = ce_ah;
  }
  return bp.used();
}

int check_bound(const kmb_handle* h) {
  if (!h->P || !h->G || !h->PB) return fail("arenas are not bound (kmb_bind_arenas)");
  if (!h->ws) return fail("workspace is not bound (kmb_bind_workspace)");
  return 0;
}

// ------------------------------------------------------------------ shared sub-graphs
struct AttnIO { const bf16_t* q; int ldq; const bf16_t* k; const bf16_t* v; int ldkv; int Tq, Tk; const int64_t* mask; int causal; };

int attn_forward(kmb_handle* h, const AttnIO& io, int B, int H, bf16_t* o, float* lse, hipStream_t s) {
  KmbAttn a; memset(&a, 0, sizeof(a));
  a.Q = io.q; a.K = io.k; a.V = io.v; a.ldq = io.ldq; a.ldk = io.ldkv; a.ldv = io.ldkv;
  a.B = B; a.H = H; a.Tq = io.Tq; a.Tk = io.Tk; a.key_mask = io.mask; a.causal = io.causal;
  a.O = o; a.ldo = h->d; a.lse = lse;
  if (g_f32) { HIPCHK(kmb_f32_attn_fwd_launch(a, s)); return 0; }
  const char* why = kmb_attn_check(a, 0);
  if (why) return fail("%s", why);
  HIPCHK(kmb_attn_fwd_launch(a, s));
  return 0;
}

int ln_forward(const bf16_t* z, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd, int M, int D,
               float eps, hipStream_t s) {
  if (g_f32) HIPCHK(kmb_f32_ln_fwd_launch((const float*)z, gamma, beta, (float*)y, mean, rstd, M, D, eps, s));
  else HIPCHK(kmb_ln_fwd_launch(z, gamma, beta, y, mean, rstd, M, D, eps, s));
  return 0;
}

int embed_ln_forward(const int64_t* ids, const int32_t* img_src, const float* E, const float* img_emb, const float* P,
                     int pos_base, int S, float scale, const float* gamma, const float* beta, bf16_t* z, bf16_t* y,
                     float* mean, float* rstd, int M, int D, float eps, KmbDrop drop, hipStream_t s) {
  if (g_f32) {
    if (drop.thr16) return fail("fp32 validation mode runs without dropout");
    HIPCHK(kmb_f32_embed_ln_fwd_launch(ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, (float*)z, (float*)y,
                                       mean, rstd, M, D, eps, s));
  } else {
    HIPCHK(kmb_embed_ln_fwd_launch(ids, img_src, E, img_emb, P, pos_base, S, scale, gamma, beta, z, y, mean, rstd, M, D,
                                   eps, drop, s));
  }
  return 0;
}

int attn_backward(kmb_handle* h, const AttnIO& io, int B, int H, bf16_t* o, float* lse, const bf16_t* dO, bf16_t* dq,
                  int lddq, bf16_t* dk, bf16_t* dv, int lddkv, float* cs_q, float* cs_k, float* cs_v, int ld_cs,
                  hipStream_t s) {
  KmbAttn a; memset(&a, 0, sizeof(a));
  a.Q = io.q; a.K = io.k; a.V = io.v; a.ldq = io.ldq; a.ldk = io.ldkv; a.ldv = io.ldkv;
  a.B = B; a.H = H; a.Tq = io.Tq; a.Tk = io.Tk; a.key_mask = io.mask; a.causal = io.causal;
  a.O = o; a.ldo = h->d; a.lse = lse; a.dO = dO; a.lddo = h->d;
  a.dQ = dq; a.lddq = lddq; a.dK = dk; a.dV = dv; a.lddk = lddkv; a.lddv = lddkv; a.dq_scale = 0.125f;
  a.dq_colsum = cs_q; a.dk_colsum = cs_k; a.dv_colsum = cs_v; a.ld_colsum = ld_cs;
  const char* why = kmb_attn_check(a, 1);
  if (why) return fail("%s", why);
  HIPCHK(kmb_attn_bwd_launch(a, s));
  return 0;
}

// post-LN FFN block forward: z = x + drop(fc2(gelu(fc1(x)))) ; out = LN(z)
int ffn_forward(kmb_handle* h, const LayerP& L, int F, const bf16_t* x, bf16_t* u, bf16_t* hh, bf16_t* z, float* mean,
                float* rstd, bf16_t* out, int M, KmbDrop dr, hipStream_t s) {
  const int d = h->d;
  KmbGemm g = lin_fwd(x, d, h->wb(L.fc1_w), h->pf(L.fc1_b), M, F, d);
  g.act = 1; g.preact = u; g.ld_preact = F; g.out_bf16 = hh; g.ld_out_bf16 = F;
  KCHK(run_gemm(g, s));
  g = lin_fwd(hh, F, h->wb(L.fc2_w), h->pf(L.fc2_b), M, d, F);
  g.drop_thr16 = dr.thr16; g.drop_seed = dr.seed; g.drop_scale = dr.scale;
  g.residual = x; g.ld_res = d; g.out_bf16 = z; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  KCHK(ln_forward(z, h->pf(L.ln_g), h->pf(L.ln_b), out, mean, rstd, M, d, h->cfg.layer_norm_eps, s));
  return 0;
}

// backward of the FFN block.  dy: grad wrt LN output.  Result: grad wrt block input x in dx_out.
int ffn_backward(kmb_handle* h, const LayerP& L, int F, const bf16_t* x, const bf16_t* u, const bf16_t* hh,
                 const bf16_t* z, const float* mean, const float* rstd, const bf16_t* dy, bf16_t* dx_out, int M,
                 KmbDrop dr, kmb_handle::BwdBufs& bb, hipStream_t s) {
  const int d = h->d;
  bf16_t* dz = bb.dz[0];
  bf16_t* dsub = dr.thr16 ? bb.dsub[0] : dz;
  KCHK(ln_backward(h, dy, z, mean, rstd, L.ln_g, L.ln_b, dz, dr.thr16 ? dsub : nullptr, KmbDrop{0u, 0u, 1.f}, dr, M, s,
                   L.fc2_b, bb.parts[0]));
  KCHK(trace("ffn.dz", dz, (size_t)M * d * 2, s));
  KCHK(wgrad_side(h, lin_wgrad(dsub, d, hh, F, h->gf(L.fc2_w), M, d, F, 0.f), s));
  KmbGemm g = lin_dgrad(dsub, d, h->wb(L.fc2_w), M, d, F);
  g.act = 2; g.aux = u; g.ld_aux = F; g.out_bf16 = bb.du; g.ld_out_bf16 = F;
  g.colsum = bb.parts[1];  // per-64-row-block column sums of du = partials of the fc1 bias gradient
  KCHK(run_gemm(g, s));
  HIPCHK(kmb_reduce_parts_launch(bb.parts[1], (M + 63) / 64, F, h->gf(L.fc1_b), F, reducer_stream(h, s)));
  KCHK(wgrad_side(h, lin_wgrad(bb.du, F, x, d, h->gf(L.fc1_w), M, F, d, 0.f), s));
  KCHK(trace("ffn.du", bb.du, (size_t)M * F * 2, s));
  g = lin_dgrad(bb.du, F, h->wb(L.fc1_w), M, F, d);
  g.residual = dz; g.ld_res = d; g.out_bf16 = dx_out; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  KCHK(trace("ffn.dx", dx_out, (size_t)M * d * 2, s));
  return 0;
}

// self-attention block forward: z = x + drop(out_proj(attn(qkv(x)))) ; out = LN(z)
int self_attn_forward(kmb_handle* h, const AttnP& A, int H, const bf16_t* x, bf16_t* qkv, bf16_t* o, float* lse,
                      bf16_t* z, float* mean, float* rstd, bf16_t* out, int B, int T, const int64_t* mask, int causal,
                      KmbDrop dr, hipStream_t s) {
  const int d = h->d, M = B * T;
  KmbGemm g = lin_fwd(x, d, h->wb(A.qkv_w), h->pf(A.qkv_b), M, 3 * d, d);
  g.col_scale = 0.125f; g.col_scale_n = d; g.out_bf16 = qkv; g.ld_out_bf16 = 3 * d;
  KCHK(run_gemm(g, s));
  AttnIO io{qkv, 3 * d, EP(qkv, d), EP(qkv, 2 * d), 3 * d, T, T, mask, causal};
  KCHK(attn_forward(h, io, B, H, o, lse, s));
  g = lin_fwd(o, d, h->wb(A.o_w), h->pf(A.o_b), M, d, d);
  g.drop_thr16 = dr.thr16; g.drop_seed = dr.seed; g.drop_scale = dr.scale;
  g.residual = x; g.ld_res = d; g.out_bf16 = z; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  KCHK(ln_forward(z, h->pf(A.ln_g), h->pf(A.ln_b), out, mean, rstd, M, d, h->cfg.layer_norm_eps, s));
  return 0;
}

int self_attn_backward(kmb_handle* h, const AttnP& A, int H, const bf16_t* x, bf16_t* qkv, bf16_t* o, float* lse,
                       const bf16_t* z, const float* mean, const float* rstd, const bf16_t* dy, bf16_t* dx_out, int B,
                       int T, const int64_t* mask, int causal, KmbDrop dr, kmb_handle::BwdBufs& bb, hipStream_t s) {
  const int d = h->d, M = B * T;
  bf16_t* dz = bb.dz[1];
  bf16_t* dsub = dr.thr16 ? bb.dsub[1] : dz;
  KCHK(ln_backward(h, dy, z, mean, rstd, A.ln_g, A.ln_b, dz, dr.thr16 ? dsub : nullptr, KmbDrop{0u, 0u, 1.f}, dr, M, s,
                   A.o_b, bb.parts[2]));
  KCHK(trace("sa.dz", dz, (size_t)M * d * 2, s));
  KCHK(wgrad_side(h, lin_wgrad(dsub, d, o, d, h->gf(A.o_w), M, d, d, 0.f), s));
  KmbGemm g = lin_dgrad(dsub, d, h->wb(A.o_w), M, d, d);
  g.out_bf16 = h->dob; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  KCHK(trace("sa.dob", h->dob, (size_t)M * d * 2, s));
  KCHK(trace("sa.o(saved)", o, (size_t)M * d * 2, s));
  KCHK(trace("sa.qkv(saved)", qkv, (size_t)M * 3 * d * 2, s));
  AttnIO io{qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, T, T, mask, causal};
  KCHK(attn_backward(h, io, B, H, o, lse, h->dob, bb.dqkv, 3 * d, bb.dqkv + d, bb.dqkv + 2 * d, 3 * d, bb.parts[3],
                     bb.parts[3] + d, bb.parts[3] + 2 * d, 3 * d, s));
  HIPCHK(kmb_reduce_parts_launch(bb.parts[3], B, 3 * d, h->gf(A.qkv_b), 3 * d, reducer_stream(h, s)));
  KCHK(trace("sa.dqkv", bb.dqkv, (size_t)M * 3 * d * 2, s));
  KCHK(wgrad_side(h, lin_wgrad(bb.dqkv, 3 * d, x, d, h->gf(A.qkv_w), M, 3 * d, d, 0.f), s));
  g = lin_dgrad(bb.dqkv, 3 * d, h->wb(A.qkv_w), M, 3 * d, d);
  g.residual = dz; g.ld_res = d; g.out_bf16 = dx_out; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  KCHK(trace("sa.dx", dx_out, (size_t)M * d * 2, s));
  return 0;
}


// One BartClassificationHead (dense -> tanh -> out_proj, classif_dropout = 0) on gathered decoder rows, its loss
// and, with need_grad, all of its gradients: parameter gradients go to the arena, the gradient wrt the decoder
// states is scatter-added (fp32) into h->dhead.  Reference src/model/model.py:133-158, :248-289.
int head_run(kmb_handle* h, int k, const bf16_t* hdec, int n, const int32_t* rows_a, const int32_t* rows_b,
             const float* soft_targets, const int64_t* labels, float factor, bool need_grad, float* loss_out,
             hipStream_t s) {
  const HeadP& H = h->head[k];
  const int d = h->d, din = H.d_in, C = H.C;
  const int Cpad = (int)align_up((size_t)C, 8);
  if (n <= 0) {  // the reference skips the term; its parameters get no gradient this step
    HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), s));
    if (need_grad) HIPCHK(hipMemsetAsync(h->gf(H.dw), 0, (H.ob + (size_t)C - H.dw) * sizeof(float), s));
    return 0;
  }
  if (n > h->head_rows_cap || h->hx == nullptr) return fail("head rows %d exceed the reserved %d (kmb_reserve_head_rows)", n, h->head_rows_cap);
  // gather: [n, d] (or [n, 2d] = object | subject for the relation head)
  HIPCHK(kmb_gather_rows_bf16_launch(hdec, d, rows_a, h->hx, din, n, d, s));
  if (rows_b) HIPCHK(kmb_gather_rows_bf16_launch(hdec, d, rows_b, h->hx + d, din, n, d, s));
  KmbGemm g = lin_fwd(h->hx, din, h->wb(H.dw), h->pf(H.db), n, d, din);
  g.act = 3; g.out_bf16 = h->hy; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  g = lin_fwd(h->hy, d, h->wb(H.ow), h->pf(H.ob), n, C, d);
  g.out_f32 = h->hlg; g.ld_out_f32 = Cpad;
  KCHK(run_gemm(g, s));
  if (soft_targets) {  // F.kl_div(log_softmax(pred), target, reduction='batchmean') * factor
    HIPCHK(kmb_kl_div_launch(h->hlg, Cpad, C, soft_targets, C, n, factor, h->hloss, need_grad ? h->hdlg : nullptr, Cpad, s));
    HIPCHK(kmb_mean_rows_launch(h->hloss, n, factor, (float)n, loss_out, s));
  } else {             // CrossEntropyLoss()(pred, labels) * factor
    HIPCHK(kmb_count_valid_launch(labels, n, C, h->count + 1, h->status, s));
    HIPCHK(kmb_ce_launch(h->hlg, Cpad, C, labels, n, h->count + 1, factor, h->hloss, need_grad ? h->hdlg : nullptr, s));
    HIPCHK(kmb_mean_rows_launch(h->hloss, n, factor, (float)n, loss_out, s));
  }
  if (!need_grad) return 0;
  // out_proj
  KCHK(bias_grad(h, h->hdlg, Cpad, n, C, h->gf(H.ob), s));
  KCHK(run_wgrad(h, lin_wgrad(h->hdlg, Cpad, h->hy, d, h->gf(H.ow), n, C, d, 0.f), s, h->head_slab, h->head_slab_floats));
  g = lin_dgrad(h->hdlg, Cpad, h->wb(H.ow), n, Cpad, d);   // reduction over the padded class dim (pad columns are zero)
  g.N = d; g.K = Cpad; g.act = 4; g.aux = h->hy; g.ld_aux = d; g.out_bf16 = h->hdy; g.ld_out_bf16 = d;
  KCHK(run_gemm(g, s));
  // dense
  KCHK(bias_grad(h, h->hdy, d, n, d, h->gf(H.db), s));
  KCHK(run_wgrad(h, lin_wgrad(h->hdy, d, h->hx, din, h->gf(H.dw), n, d, din, 0.f), s, h->head_slab, h->head_slab_floats));
  g = lin_dgrad(h->hdy, d, h->wb(H.dw), n, d, din);
  g.out_bf16 = h->hdx; g.ld_out_bf16 = din;
  KCHK(run_gemm(g, s));
  HIPCHK(kmb_scatter_add_rows_launch(h->hdx, din, rows_a, h->dhead, n, d, s));
  if (rows_b) HIPCHK(kmb_scatter_add_rows_launch(h->hdx + d, din, rows_b, h->dhead, n, d, s));
  return 0;
}

int encoder_forward(kmb_handle* h, const kmb_batch& bt, bool train, hipStream_t s) {
  const int d = h->d, B = bt.B, S = bt.S, Me = B * S;
  const float eps = h->cfg.layer_norm_eps;
  const float scale = h->cfg.scale_embedding ? sqrtf((float)d) : 1.f;
  if (bt.n_features > 0 && g_f32) {   // the raw fp32 features against the fp32 master weight [d, Fin]
    KmbGemm g = lin_fwd(reinterpret_cast<const bf16_t*>(bt.image_features), h->Fin, h->wb(h->img_w), h->pf(h->img_b),
                        bt.n_features, d, h->Fin);
    g.out_f32 = h->img_emb; g.ld_out_f32 = d;
    KCHK(run_gemm(g, s));
  } else if (bt.n_features > 0) {
    HIPCHK(kmb_cast_pad_launch(bt.image_features, bt.n_features, h->Fin, h->xf, h->Fpad, s));
    KmbGemm g = lin_fwd(h->xf, h->Fpad, h->imgw_pad, h->pf(h->img_b), bt.n_features, d, h->Fpad);
    g.out_f32 = h->img_emb; g.ld_out_f32 = d;
    KCHK(run_gemm(g, s));
  }
  HIPCHK(kmb_img_rowmap_launch(bt.input_ids, bt.feat_offsets, B, S, h->cfg.img_feat_id, h->cfg.cls_token_id,
                               h->img_src, h->status, s));
  KCHK(embed_ln_forward(bt.input_ids, h->img_src, h->pf(h->shared), h->img_emb, h->pf(h->enc_pos),
                        h->cfg.extra_pos_embeddings, S, scale, h->pf(h->enc_lne_g), h->pf(h->enc_lne_b),
                        h->ze0, h->xe[0], h->me0, h->re0, Me, d, eps, h->drop_site(1, train), s));
  for (int l = 0; l < h->cfg.encoder_layers; ++l) {
    const LayerP& L = h->enc[l];
    EncAct& a = h->ea[l];
    KCHK(self_attn_forward(h, L.sa, h->He, h->xe[l], a.qkv, a.o, a.lse, a.z1, a.m1, a.r1, a.y1, B, S,
                           bt.attention_mask, 0, h->drop_site(10 + 2 * l, train), s));
    KCHK(ffn_forward(h, L, h->Fe, a.y1, a.u, a.hh, a.z2, a.m2, a.r2, h->xe[l + 1], Me, h->drop_site(11 + 2 * l, train), s));
  }
  return 0;
}

}  // namespace

// =============================================================================================
int kmb_set_error(const char* msg) { g_err = msg ? msg : ""; return 1; }

extern "C" {

const char* kmb_last_error(void) { return g_err.c_str(); }
int kmb_version(void) { return 1; }

int kmb_create(const kmb_config* cfg, kmb_handle** out) {
  if (!cfg || !out) return fail("kmb_create: null argument");
  if (cfg->d_model != cfg->encoder_attention_heads * 64 || cfg->d_model != cfg->decoder_attention_heads * 64)
    return fail("kmb_create: head_dim must be 64 (d_model=%d heads=%d/%d)", cfg->d_model, cfg->encoder_attention_heads,
                cfg->decoder_attention_heads);
  if (cfg->d_model > 2048) return fail("kmb_create: d_model > 2048 is not supported");
  if ((cfg->encoder_ffn_dim & 63) || (cfg->decoder_ffn_dim & 63)) return fail("kmb_create: ffn dims must be multiples of 64");
  if (cfg->attention_dropout != 0.f || cfg->activation_dropout != 0.f)
    return fail("kmb_create: attention_dropout / activation_dropout != 0 are not implemented (vcg_base uses 0.0)");
  kmb_handle* h = new kmb_handle();
  h->cfg = *cfg;
  if (h->cfg.layer_norm_eps <= 0.f) h->cfg.layer_norm_eps = 1e-5f;
  h->d = cfg->d_model; h->He = cfg->encoder_attention_heads; h->Hd = cfg->decoder_attention_heads;
  h->Fe = cfg->encoder_ffn_dim; h->Fd = cfg->decoder_ffn_dim; h->V = cfg->vocab_size;
  h->Vpad = (int)align_up((size_t)cfg->vocab_size, 128);
  h->Fin = cfg->image_feature_size; h->Fpad = (int)align_up((size_t)cfg->image_feature_size, 64);   // K of the image projection: a multiple of the GEMMs' K step
  h->Prows = cfg->max_position_embeddings + cfg->extra_pos_embeddings;
  const int d = h->d;
  h->img_w = add_param(h, "model.encoder.embed_images.linear.weight", d, h->Fin);
  h->img_b = add_param(h, "model.encoder.embed_images.linear.bias", 1, d);
  h->enc_pos = add_param(h, "model.encoder.embed_positions.weight", h->Prows, d);
  h->enc_lne_g = add_param(h, "model.encoder.layernorm_embedding.weight", 1, d);
  h->enc_lne_b = add_param(h, "model.encoder.layernorm_embedding.bias", 1, d);
  std::vector<size_t> marks;  // bucket boundaries in arena order
  h->enc.resize(cfg->encoder_layers);
  for (int l = 0; l < cfg->encoder_layers; ++l) {
    marks.push_back(align_up(h->arena, 64));
    const std::string p = "model.encoder.layers." + std::to_string(l) + ".";
    add_attn(h, p + "self_attn.", p + "self_attn_layer_norm", h->enc[l].sa);
    add_ffn(h, p, h->Fe, h->enc[l]);
  }
  marks.push_back(align_up(h->arena, 64));
  h->dec_pos = add_param(h, "model.decoder.embed_positions.weight", h->Prows, d);
  h->dec_lne_g = add_param(h, "model.decoder.layernorm_embedding.weight", 1, d);
  h->dec_lne_b = add_param(h, "model.decoder.layernorm_embedding.bias", 1, d);
  h->dec.resize(cfg->decoder_layers);
  // every layer's cross-attention k | v weights, then their biases: part of the decoder-embedding segment of the arena
  // (its gradients are complete when the batched weight gradient behind the last decoder layer's backward has run)
  h->xkv_w = align_up(h->arena, 64);
  for (int l = 0; l < cfg->decoder_layers; ++l) {
    const std::string p = "model.decoder.layers." + std::to_string(l) + ".encoder_attn.";
    h->dec[l].ca_kv_w = add_param(h, p + "k_proj.weight", d, d);
    add_param(h, p + "v_proj.weight", d, d);
  }
  h->xkv_b = align_up(h->arena, 64);
  for (int l = 0; l < cfg->decoder_layers; ++l) {
    const std::string p = "model.decoder.layers." + std::to_string(l) + ".encoder_attn.";
    h->dec[l].ca_kv_b = add_param(h, p + "k_proj.bias", 1, d);
    add_param(h, p + "v_proj.bias", 1, d);
  }
  for (int l = 0; l < cfg->decoder_layers; ++l) {
    marks.push_back(align_up(h->arena, 64));
    const std::string p = "model.decoder.layers." + std::to_string(l) + ".";
    add_attn(h, p + "self_attn.", p + "self_attn_layer_norm", h->dec[l].sa);
    add_cross_attn(h, p + "encoder_attn.", p + "encoder_attn_layer_norm", h->dec[l].ca);
    add_ffn(h, p, h->Fd, h->dec[l]);
  }
  marks.push_back(align_up(h->arena, 64));
  {  // BartClassificationHead: dense -> tanh -> out_proj
    const char* names[3] = {"mrm_head", "attribute_head", "relation_head"};
    const int C[3] = {cfg->num_labels, cfg->num_attributes, cfg->num_relations};
    h->heads_begin = align_up(h->arena, 64);
    for (int k = 0; k < 3; ++k) {
      if (C[k] <= 0) continue;
      HeadP& H = h->head[k];
      H.on = true; H.C = C[k]; H.d_in = (k == 2) ? 2 * d : d;
      const std::string n = names[k];
      H.dw = add_param(h, n + ".dense.weight", d, H.d_in);
      H.db = add_param(h, n + ".dense.bias", 1, d);
      H.ow = add_param(h, n + ".out_proj.weight", H.C, d);
      H.ob = add_param(h, n + ".out_proj.bias", 1, H.C);
    }
    h->heads_end = align_up(h->arena, 64);
  }
  h->shared = add_param(h, "model.shared.weight", h->V, d);
  h->arena = align_up(h->arena, 64);
  // arena segments: [0,m0) enc embed+img | enc layers | dec embed | dec layers | shared
  // completion order in backward: dec layers (last..first), dec embed, enc layers (last..first), enc embed, shared
  const int Le = cfg->encoder_layers, Ld = cfg->decoder_layers;
  auto seg = [&](size_t a, size_t b) { h->buckets.push_back({a, b - a}); };
  for (int l = Ld - 1; l >= 0; --l) seg(marks[Le + 1 + l], marks[Le + 2 + l]);
  // (the decoder-layer-0 segment ends where the heads begin; the heads' gradients are complete after forward and
  //  ride in the last bucket together with the tied matrix)
  seg(marks[Le], marks[Le + 1]);
  for (int l = Le - 1; l >= 0; --l) seg(marks[l], marks[l + 1]);
  seg(0, marks[0]);
  seg(h->heads_begin, h->arena);  // heads (if any) + tied matrix
  h->events.resize(h->buckets.size());
  for (auto& e : h->events) {
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      // no device (CPU build container): events are created lazily in kmb_backward instead
      e = nullptr;
      (void)hipGetLastError();
    }
  }
  const char* ch = KMB_DIAG_ENV("KMB_LM_CHUNK");
  if (ch && atoi(ch) > 0) h->lm_chunk = atoi(ch);
  *out = h;
  return 0;
}

void kmb_destroy(kmb_handle* h) {
  if (!h) return;
  for (auto e : h->events) if (e) (void)hipEventDestroy(e);
  for (auto e : h->ring) if (e) (void)hipEventDestroy(e);
  for (auto e : h->layer_done) if (e) (void)hipEventDestroy(e);
  if (h->head_wgrad_done) (void)hipEventDestroy(h->head_wgrad_done);
  if (h->side) (void)hipStreamDestroy(h->side);
  (void)kmb_comm_destroy(h);
  delete h;
}

int kmb_param_count(const kmb_handle* h) { return (int)h->params.size(); }
int kmb_param_info(const kmb_handle* h, int idx, const char** name, int64_t* offset, int32_t* rows, int32_t* cols) {
  if (idx < 0 || idx >= (int)h->params.size()) return fail("kmb_param_info: index %d out of range", idx);
  const ParamInfo& p = h->params[idx];
  if (name) *name = p.name.c_str();
  if (offset) *offset = (int64_t)p.off;
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  return 0;
}
int64_t kmb_arena_elems(const kmb_handle* h) { return (int64_t)h->arena; }
int64_t kmb_bf16_arena_elems(const kmb_handle* h) {
  // mirror + zero rows padding the tied matrix to Vpad + padded image weight [d, Fpad]
  return (int64_t)(h->shared + (size_t)h->Vpad * h->d + (size_t)h->d * h->Fpad + 64);
}
int kmb_logits_ld(const kmb_handle* h) { return h->Vpad; }

int kmb_bind_arenas(kmb_handle* h, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                    kmb_bf16* params_bf16, float* final_logits_bias) {
  if (!params || !params_bf16 || !final_logits_bias) return fail("kmb_bind_arenas: params, bf16 mirror and final_logits_bias are required");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)params_bf16) & 255)
    return fail("kmb_bind_arenas: arenas must be 256-byte aligned");
  h->P = params; h->G = grads; h->M1 = exp_avg; h->M2 = exp_avg_sq; h->PB = params_bf16; h->flb = final_logits_bias;
  h->imgw_pad = h->PB + align_up(h->shared + (size_t)h->Vpad * h->d, 64);
  h->mirror_version++;
  return 0;
}

int64_t kmb_workspace_bytes(const kmb_handle* h, int B, int S, int T, int n_features) {
  PrecisionScope scope(h);
  return (int64_t)layout_train(const_cast<kmb_handle*>(h), nullptr, 0, B, S, T, n_features, false);
}
int kmb_bind_workspace(kmb_handle* h, void* ws, int64_t bytes) {
  if (((uintptr_t)ws) & 255) return fail("kmb_bind_workspace: workspace must be 256-byte aligned");
  h->ws = (char*)ws; h->ws_bytes = (size_t)bytes; h->have_fwd = false; h->have_hdec = false; h->gen.active = false; h->gen.packed_at = nullptr;
  return 0;
}

int kmb_set_seed(kmb_handle* h, uint64_t seed) { h->seed = seed; h->step = 0; return 0; }

int kmb_sync_params(kmb_handle* h, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h->P || !h->PB) return fail("kmb_sync_params: arenas are not bound");
  HIPCHK(kmb_cast_f32_bf16_launch(h->P, h->PB, h->arena, s));
  h->mirror_version++;
  // zero rows [V, Vpad) of the tied matrix mirror; padded image weight [d, Fpad]
  HIPCHK(hipMemsetAsync(h->PB + h->shared + (size_t)h->V * h->d, 0, (size_t)(h->Vpad - h->V) * h->d * sizeof(bf16_t), s));
  HIPCHK(kmb_cast_rows_launch(h->pf(h->img_w), h->Fin, h->imgw_pad, h->Fpad, h->d, h->Fin, s));
  return 0;
}

int kmb_bucket_count(const kmb_handle* h) { return (int)h->buckets.size(); }
int kmb_bucket_range(const kmb_handle* h, int i, int64_t* offset, int64_t* count) {
  if (i < 0 || i >= (int)h->buckets.size()) return fail("kmb_bucket_range: index out of range");
  *offset = (int64_t)h->buckets[i].off; *count = (int64_t)h->buckets[i].count;
  return 0;
}
int kmb_stream_wait_bucket(kmb_handle* h, int i, void* stream) {
  if (i < 0 || i >= (int)h->events.size() || !h->events[i]) return fail("kmb_stream_wait_bucket: no event %d", i);
  HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->events[i], 0));
  return 0;
}

int kmb_read_status(kmb_handle* h, int32_t* status_host, void* stream) {
  if (!h->status) return fail("kmb_read_status: no forward has run");
  HIPCHK(hipMemcpyAsync(status_host, h->status, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int kmb_read_status_async(kmb_handle* h, int32_t* status_host, void* stream) {
  if (!h->status) return fail("kmb_read_status_async: no forward has run");
  HIPCHK(hipMemcpyAsync(status_host, h->status, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

// --------------------------------------------------------------------------------- forward
// the tied LM head on rows [0, Md) of hdec: fp32 logits [Md, Vpad] (src/model/model.py:397)
// The vocabulary projection with fp32 logits.  A decode step's rows (batch x beams = 257 .. 320: the benchmarked 64 x 5): one
// workgroup per 256 vocabulary columns holds ALL rows, so every row of the tied matrix crosses a CU's memory pipe once
// (gemm.hip "All rows" kernel: 58 -> 47 us at 320 rows, tools/allrows_time.py; at <= 192 rows the 128x128 tiles already read
// the matrix once or twice and are faster: 31 against 39 us).  Bit-identical either way; KMB_GEMM_ALLROWS=0: always the tuner's pick.
// stats / stats_blocks (a generation step): the all-rows kernel also leaves every row's per-block (maximum, sum-exp) pairs there and
// *stats_blocks = their number, for the beam step that follows (loss.hip beam_stats_merge_kernel); 0 when another kernel ran.
static int run_vocab_gemm(const KmbGemm& g, hipStream_t s, float* stats = nullptr, int* stats_blocks = nullptr) {
  static const bool allrows_ok = !(KMB_DIAG_ENV("KMB_GEMM_ALLROWS") && KMB_DIAG_ENV("KMB_GEMM_ALLROWS")[0] == '0');
  if (stats_blocks) *stats_blocks = 0;
  if (allrows_ok && !g_f32 && g.M > 256 && g.M <= 320 && kmb_gemm_allrows_check(g) == nullptr) {
    HIPCHK(kmb_gemm_allrows_launch(g, stats_blocks ? stats : nullptr, s));
    if (stats_blocks && stats) *stats_blocks = kmb_gemm_allrows_blocks(g.N);
    return 0;
  }
  return run_gemm(g, s);
}

static int head_logits(kmb_handle* h, const bf16_t* hdec, int Md, float* logits_out, hipStream_t s) {
  const int d = h->d;
  const int CH = Md < h->lm_chunk ? Md : h->lm_chunk;
  for (int r0 = 0; r0 < Md; r0 += CH) {
    const int rows = (Md - r0) < CH ? (Md - r0) : CH;
    KmbGemm g = lin_fwd(EP(hdec, (size_t)r0 * d), d, h->wb(h->shared), h->flb, rows, h->V, d);
    g.out_f32 = logits_out + (size_t)r0 * h->Vpad; g.ld_out_f32 = h->Vpad;
    KCHK(run_vocab_gemm(g, s));
  }
  return 0;
}

static int forward_impl(kmb_handle* h, const kmb_batch* batch, const kmb_pretrain* extra, const kmb_forward_opts* opts,
                        int train, int need_grad, float* loss_out, float* logits_out, kmb_bf16* enc_out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  KCHK(check_bound(h));
  PrecisionScope scope(h);
  if (h->fp32 && (train || need_grad || extra))
    return fail("kmb_forward: the fp32 validation mode (kmb_set_precision) is an eval-mode forward only");
  const bf16_t* enc_in = opts ? opts->encoder_states : nullptr;
  // need_grad with given encoder states: backward stops at them (kmb_encoder_states_grad hands out dL/d states, the encoder's
  // parameter gradients are zero) -- src/model/model.py:76-83 run with a tensor that requires grad
  if (!batch || !batch->input_ids || !batch->decoder_input_ids || !batch->feat_offsets)
    return fail("kmb_forward: input_ids, decoder_input_ids and feat_offsets are required");
  const kmb_batch& bt = *batch;
  if (bt.B <= 0 || bt.S <= 0 || bt.T <= 0) return fail("kmb_forward: empty batch");
  if (bt.S > h->cfg.max_position_embeddings || bt.T > h->cfg.max_position_embeddings)
    return fail("kmb_forward: sequence longer than max_position_embeddings");
  if (need_grad && !bt.labels && !extra) return fail("kmb_forward: need_grad requires labels");
  if (need_grad && (!h->G)) return fail("kmb_forward: gradient arena is not bound");
  const size_t need = layout_train(h, nullptr, 0, bt.B, bt.S, bt.T, bt.n_features, false);
  if (need > h->ws_bytes) return fail("kmb_forward: workspace too small (%zu > %zu bytes)", need, h->ws_bytes);
  layout_train(h, h->ws, h->ws_bytes, bt.B, bt.S, bt.T, bt.n_features, true);
  if (!h->fp32 && h->small_floats > 1024) { g_small_slab = h->small_slab; g_small_floats = h->small_floats; }
  h->gen.active = false; h->gen.packed_at = nullptr;   // the activations of this forward overwrite the generation workspace
  const int d = h->d, B = bt.B, S = bt.S, T = bt.T, Me = B * S, Md = B * T;
  h->bt = bt; h->Me = Me; h->Md = Md; h->Ntot = bt.n_features;
  h->fwd_train = train != 0; h->have_fwd = false; h->have_hdec = false; h->have_bwd = false;
  h->enc_given = enc_in != nullptr;
  if (train) h->step += 1;
  const bool tr = train != 0;
  const float eps = h->cfg.layer_norm_eps;
  const float scale = h->cfg.scale_embedding ? sqrtf((float)d) : 1.f;
  HIPCHK(hipMemsetAsync(h->status, 0, 16, s));

  // encoder once, unless the caller already holds its output (src/model/model.py:76-83)
  if (enc_in) HIPCHK(hipMemcpyAsync(h->xe[h->cfg.encoder_layers], enc_in, (size_t)Me * d * esz(), hipMemcpyDeviceToDevice, s));
  else KCHK(encoder_forward(h, bt, tr, s));
  const bf16_t* enc = h->xe[h->cfg.encoder_layers];
  if (enc_out) HIPCHK(hipMemcpyAsync(enc_out, enc, (size_t)Me * d * esz(), hipMemcpyDeviceToDevice, s));

  // ---- decoder (teacher forced): HF3.0.2 BartDecoder.forward via src/model/model.py:87-97
  KCHK(embed_ln_forward(bt.decoder_input_ids, nullptr, h->pf(h->shared), nullptr, h->pf(h->dec_pos),
                        h->cfg.extra_pos_embeddings, T, scale, h->pf(h->dec_lne_g), h->pf(h->dec_lne_b),
                        h->zd0, h->xd[0], h->md0, h->rd0, Md, d, eps, h->drop_site(2, tr), s));
  const int Ldec = h->cfg.decoder_layers;
  const int ldkv = Ldec * 2 * d;   // row stride of the batched cross-attention k | v buffers
  if (Ldec > 0) {   // every decoder layer's cross-attention keys | values in ONE GEMM (same input: the encoder output)
    KmbGemm g = lin_fwd(enc, d, h->wb(h->xkv_w), h->pf(h->xkv_b), Me, ldkv, d);
    g.out_bf16 = h->ckv_all; g.ld_out_bf16 = ldkv;
    KCHK(run_gemm(g, s));
  }
  for (int l = 0; l < h->cfg.decoder_layers; ++l) {
    const LayerP& L = h->dec[l];
    DecAct& a = h->da[l];
    KCHK(self_attn_forward(h, L.sa, h->Hd, h->xd[l], a.qkv, a.o1, a.lse1, a.z1, a.m1, a.r1, a.y1, B, T,
                           bt.decoder_attention_mask, 1, h->drop_site(100 + 3 * l, tr), s));
    // cross attention: q from decoder states (scaled), k|v from the encoder output
    KmbGemm g = lin_fwd(a.y1, d, h->wb(L.ca.qkv_w), h->pf(L.ca.qkv_b), Md, d, d);
    g.col_scale = 0.125f; g.col_scale_n = d; g.out_bf16 = a.cq; g.ld_out_bf16 = d;
    KCHK(run_gemm(g, s));
    AttnIO io{a.cq, d, a.ckv, EP(a.ckv, d), ldkv, T, S, bt.attention_mask, 0};
    KCHK(attn_forward(h, io, B, h->Hd, a.o2, a.lse2, s));
    const KmbDrop dr = h->drop_site(101 + 3 * l, tr);
    g = lin_fwd(a.o2, d, h->wb(L.ca.o_w), h->pf(L.ca.o_b), Md, d, d);
    g.drop_thr16 = dr.thr16; g.drop_seed = dr.seed; g.drop_scale = dr.scale;
    g.residual = a.y1; g.ld_res = d; g.out_bf16 = a.z2; g.ld_out_bf16 = d;
    KCHK(run_gemm(g, s));
    KCHK(ln_forward(a.z2, h->pf(L.ca.ln_g), h->pf(L.ca.ln_b), a.y2, a.m2, a.r2, Md, d, eps, s));
    KCHK(ffn_forward(h, L, h->Fd, a.y2, a.u, a.hh, a.z3, a.m3, a.r3, h->xd[l + 1], Md, h->drop_site(102 + 3 * l, tr), s));
  }
  const bf16_t* hdec = h->xd[h->cfg.decoder_layers];
  h->have_hdec = true;
  if (opts && opts->decoder_states_out)   // MultiModalBartModel.forward returns the decoder states (src/model/model.py:100-103)
    HIPCHK(hipMemcpyAsync(opts->decoder_states_out, hdec, (size_t)Md * d * esz(), hipMemcpyDeviceToDevice, s));
  if (opts && opts->skip_head) return 0;

  // ---- tied LM head + CE (src/model/model.py:397-403), in row chunks of lm_chunk (8192) rows: one launch at the
  // benchmark batch.  Smaller chunks (KMB_LM_CHUNK) keep the fp32 logits on-die but quantise the tile count worse:
  // 512-row chunks measured 0.8 % slower end to end.
  if (bt.labels) HIPCHK(kmb_count_valid_launch(bt.labels, Md, h->V, h->count, h->status, s));
  if (bt.labels || logits_out) {
    // bf16 head (product path, no logits requested): ONE GEMM writes bf16 logits into dlogits_c and the CE kernel turns
    // them into the gradient in place.  fp32 head: logits requested by the caller / fp32 validation mode / very wide
    // vocabularies, in row chunks that bound the fp32 buffer.
    // KMB_FP32_HEAD=1 forces the fp32 head (chunked fp32 logits + ce_kernel_reg) in the product mode as well: the bf16 head
    // rounds logits of magnitude 10-20 to 8 significant bits before the softmax (the reference's AMP path holds fp16
    // logits, its CPU path fp32); measured effect on the vcg_base loss 5e-5 relative either way.
    const bool bf16_head = !g_f32 && !logits_out && h->Vpad <= 65536 && !force_fp32_head();
    const int CH = bf16_head ? Md : (Md < h->lm_chunk ? Md : h->lm_chunk);
    const bf16_t* Eb = h->wb(h->shared);
    const float lmf = extra ? extra->lm_factor : 1.f;
    // Cross-entropy without a pass over the logits (loss.hip "Tied-head cross-entropy WITHOUT a pass over the logits"):
    // the head GEMM stores exp(logit - label's logit) and per-row sums, the data- and weight-gradient GEMMs run on that
    // matrix with per-row factors applied outside.  Needs whole 256-row / 256-column tiles (Md % 256 == 0; the vocabulary is
    // padded to Vpad with a -1e30 bias); other shapes, fp32 logits and KMB_FUSED_CE=0 take the two-kernel path below.
    const char* fce = getenv("KMB_FUSED_CE");     // read per call: tests flip it inside one process
    const bool fused_ce_env = !(fce && fce[0] == '0');
    const int nparts = h->Vpad / 64;
    const bool fused_ce = bf16_head && bt.labels && fused_ce_env && (Md % 256) == 0 && (h->Vpad % 256) == 0 && (d % 64) == 0 &&
                          d >= 128 && (long)(Md / 256) * (h->Vpad / 256) >= 128 && (size_t)Md * nparts <= h->logits_c_floats;
    if (fused_ce) {
      HIPCHK(kmb_ce_label_logit_launch(hdec, d, Eb, d, h->flb, bt.labels, Md, d, h->V, h->ce_shift, s));
      HIPCHK(kmb_ce_pad_bias_launch(h->flb, h->V, h->Vpad, h->ce_bias, s));
      KmbGemm g = lin_fwd(hdec, d, Eb, h->ce_bias, Md, h->Vpad, d);
      g.act = 5; g.out_bf16 = h->dlogits_c; g.ld_out_bf16 = h->Vpad;
      g.row_shift = h->ce_shift; g.row_sums = h->logits_c; g.row_sums_ld = nparts; g.pick_col = nullptr; g.pick_out = nullptr;   // the shift is the label's logit itself: v[label] - shift = 0 up to summation order
      KCHK(run_gemm(g, s));
      HIPCHK(kmb_ce_rows_finish_launch(h->logits_c, nparts, nparts, nullptr, bt.labels, h->count, lmf, Md, d, h->V, hdec, d,
                                       h->loss_rows, h->ce_srow, h->ce_alpha, need_grad ? h->ce_ah : nullptr,
                                       need_grad ? h->dlogits_c : nullptr, h->Vpad, s));
      if (need_grad) {
        // dH_r = a_r sum_j P'_rj E_j: the GEMM on P' into fp32 slabs (the row sums in that buffer were consumed by the launch
        // above), then the finish
        KmbGemm gd = lin_dgrad(h->dlogits_c, h->Vpad, Eb, Md, h->Vpad, d);
        const int tiles256 = ((Md + 255) / 256) * ((d + 255) / 256);
        static const int rounds = KMB_DIAG_ENV("KMB_HEAD_DGRAD_ROUNDS") ? atoi(KMB_DIAG_ENV("KMB_HEAD_DGRAD_ROUNDS")) : 3;   // tuning knob
        int S = tiles256 > 0 ? (256 * rounds) / tiles256 : 1;
        if (S > 8) S = 8;
        while (S > 1 && (size_t)S * Md * d > h->logits_c_floats) --S;
        if (S > 1 && h->Vpad / 64 >= 2 * S) {
          gd.split_k = S; gd.slab = h->logits_c;
        } else {
          S = 1; gd.out_f32 = h->logits_c; gd.ld_out_f32 = d;
        }
        KCHK(run_gemm(gd, s));
        HIPCHK(kmb_ce_dgrad_finish_launch(h->logits_c, S, (size_t)Md * d, h->ce_alpha, h->dhdec, Md, d, s));
        // dE = P'^T (a . H), on the side stream like the two-kernel path's
        KCHK(ensure_side(h));
        KCHK(wgrad_side(h, lin_wgrad(h->dlogits_c, h->Vpad, h->ce_ah, d, h->gf(h->shared), Md, h->V, d, 0.f), s));
        if (h->side_on && h->side) {
          HIPCHK(hipEventRecord(h->head_wgrad_done, h->side));
          h->head_wgrad_pending = true;
        }
      }
    }
    for (int r0 = 0, c = 0; !fused_ce && r0 < Md; r0 += CH, ++c) {
      const int rows = (Md - r0) < CH ? (Md - r0) : CH;
      KmbGemm g = lin_fwd(EP(hdec, (size_t)r0 * d), d, Eb, h->flb, rows, h->V, d);
      if (bf16_head) {
        bf16_t* lg = h->dlogits_c + (size_t)r0 * h->Vpad;
        g.out_bf16 = lg; g.ld_out_bf16 = h->Vpad;
        KCHK(run_gemm(g, s));
        if (bt.labels)
          HIPCHK(kmb_ce_bf16_launch(lg, h->Vpad, h->V, bt.labels + r0, rows, h->count, lmf, h->loss_rows + r0,
                                    need_grad ? lg : nullptr, s));
        continue;
      }
      float* lg = logits_out ? logits_out + (size_t)r0 * h->Vpad : h->logits_c;
      if (!logits_out && (size_t)rows * h->Vpad > h->logits_c_floats) return fail("forward: the fp32 logits chunk does not fit the workspace's logits buffer");
      g.out_f32 = lg; g.ld_out_f32 = h->Vpad;
      KCHK(run_gemm(g, s));
      if (!bt.labels) continue;
      HIPCHK(kmb_ce_launch(lg, h->Vpad, h->V, bt.labels + r0, rows, h->count, lmf, h->loss_rows + r0,
                           need_grad ? h->dlogits_c + (size_t)r0 * h->Vpad : nullptr, s));
    }
    if (bt.labels && need_grad && !fused_ce) {
      // dH = dlogits E  (reduction over the padded vocabulary; pad columns / rows are zero)
      KmbGemm gd = lin_dgrad(h->dlogits_c, h->Vpad, Eb, Md, h->Vpad, d);
      // [Md, d] has few 256x256 tiles (192 at Md = 16384: three quarters of the CUs) and the reduction runs over the
      // whole vocabulary (788 K steps): split it so that the grid is a whole number of rounds; a small pass sums the
      // slabs into the bf16 gradient.  The slabs live in the fp32 logits buffer, which is free once the CE ran.
      const int tiles256 = ((Md + 255) / 256) * ((d + 255) / 256);
      const size_t CHl = h->logits_c_floats;   // floats in the slab / logits buffer
      static const int rounds = KMB_DIAG_ENV("KMB_HEAD_DGRAD_ROUNDS") ? atoi(KMB_DIAG_ENV("KMB_HEAD_DGRAD_ROUNDS")) : 3;   // tuning knob
      int S = tiles256 > 0 ? (256 * rounds) / tiles256 : 1;
      if (S > 8) S = 8;
      while (S > 1 && (size_t)S * Md * d > CHl) --S;
      if (S > 1 && ((size_t)Md * d & 7) == 0 && h->Vpad / 64 >= 2 * S) {
        gd.split_k = S; gd.slab = h->logits_c;
        KCHK(run_gemm(gd, s));
        HIPCHK(kmb_reduce_slabs_bf16_launch(h->logits_c, S, (size_t)Md * d, h->dhdec, (size_t)Md * d, s));
      } else {
        gd.out_bf16 = h->dhdec; gd.ld_out_bf16 = d;
        KCHK(run_gemm(gd, s));
      }
      // dE[V,d] = dlogits^T H  (overwrites: the embedding scatter-adds of backward come on top)
      // (on the side stream: it overlaps the start of backward; the embedding scatter-adds wait for it)
      KCHK(ensure_side(h));
      KCHK(wgrad_side(h, lin_wgrad(h->dlogits_c, h->Vpad, hdec, d, h->gf(h->shared), Md, h->V, d, 0.f), s));
      if (h->side_on && h->side) {
        HIPCHK(hipEventRecord(h->head_wgrad_done, h->side));
        h->head_wgrad_pending = true;
      }
    }
    if (bt.labels) {
      HIPCHK(kmb_loss_finish_launch(h->loss_rows, Md, h->count, h->loss_dev, s));
      if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, h->loss_dev, sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  if (extra && need_grad && !bt.labels) {
    // no LM term (src/model/model.py:293-302 allows labels=None beside head labels): the decoder-state gradient and the
    // tied matrix's gradient start from zero; the heads add into the former, backward's embedding scatter-adds into the latter
    HIPCHK(hipMemsetAsync(h->dhdec, 0, (size_t)Md * d * sizeof(bf16_t), s));
    HIPCHK(hipMemsetAsync(h->gf(h->shared), 0, (size_t)h->V * d * sizeof(float), s));
    h->head_wgrad_pending = false;
  }
  if (extra) {
    // ---- pre-training heads on the decoder states (src/model/model.py:248-289) and the weighted total (:304-307)
    const bool any = (extra->n_mrm > 0 && h->head[0].on) || (extra->n_attr > 0 && h->head[1].on) ||
                     (extra->n_rel > 0 && h->head[2].on);
    if (need_grad && any) HIPCHK(hipMemsetAsync(h->dhead, 0, (size_t)Md * d * sizeof(float), s));
    HIPCHK(hipMemsetAsync(h->losses5, 0, 8 * sizeof(float), s));
    if (bt.labels) HIPCHK(kmb_mean_rows_launch(h->loss_dev, 1, extra->lm_factor, 1.f, h->losses5 + 1, s));
    if (h->head[0].on)
      KCHK(head_run(h, 0, hdec, extra->n_mrm, extra->mrm_rows, nullptr, extra->mrm_targets, nullptr, extra->mrm_factor,
                    need_grad != 0, h->losses5 + 2, s));
    if (h->head[1].on)
      KCHK(head_run(h, 1, hdec, extra->n_attr, extra->attr_rows, nullptr, nullptr, extra->attr_labels,
                    extra->attr_factor, need_grad != 0, h->losses5 + 3, s));
    if (h->head[2].on)
      KCHK(head_run(h, 2, hdec, extra->n_rel, extra->rel_obj_rows, extra->rel_subj_rows, nullptr, extra->rel_labels,
                    extra->rel_factor, need_grad != 0, h->losses5 + 4, s));
    if (need_grad && any) HIPCHK(kmb_add_f32_into_bf16_launch(h->dhdec, h->dhead, (size_t)Md * d, s));
    HIPCHK(kmb_mean_rows_launch(h->losses5 + 1, 4, 1.f, 1.f, h->losses5, s));
    if (extra->losses_out)
      HIPCHK(hipMemcpyAsync(extra->losses_out, h->losses5, 5 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  h->have_fwd = need_grad != 0;
  return 0;
}

int kmb_forward(kmb_handle* h, const kmb_batch* batch, int train, int need_grad, float* loss_out, float* logits_out,
                kmb_bf16* enc_out, void* stream) {
  return forward_impl(h, batch, nullptr, nullptr, train, need_grad, loss_out, logits_out, enc_out, stream);
}

int kmb_forward_ex(kmb_handle* h, const kmb_batch* batch, const kmb_forward_opts* opts, int train, int need_grad,
                   float* loss_out, float* logits_out, kmb_bf16* enc_out, void* stream) {
  return forward_impl(h, batch, nullptr, opts, train, need_grad, loss_out, logits_out, enc_out, stream);
}

int kmb_hidden_state(kmb_handle* h, int which, int index, kmb_bf16* out, void* stream) {
  if (!h->have_hdec) return fail("kmb_hidden_state: no forward whose activations are still in the workspace");
  if (h->fp32) return fail("kmb_hidden_state: not available in the fp32 validation mode");
  const auto& xs = which == 0 ? h->xe : h->xd;
  if ((which != 0 && which != 1) || index < 0 || index >= (int)xs.size() || !xs[index]) return fail("kmb_hidden_state: index out of range");
  const size_t rows = which == 0 ? (size_t)h->Me : (size_t)h->Md;
  HIPCHK(hipMemcpyAsync(out, xs[index], rows * h->d * sizeof(bf16_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// dL / d(encoder states) of the last kmb_backward as bf16 [B * S, d_model]: the sum of every decoder layer's cross-attention key /
// value data gradients, i.e. what the encoder's backward starts from (and all there is when the forward was given its states)
int kmb_encoder_states_grad(kmb_handle* h, kmb_bf16* out, void* stream) {
  if (!h || !out) return fail("kmb_encoder_states_grad: handle and out are required");
  if (!h->have_bwd) return fail("kmb_encoder_states_grad: run kmb_backward first");
  HIPCHK(hipMemcpyAsync(out, h->denc, (size_t)h->Me * h->d * sizeof(bf16_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int kmb_attention_probs(kmb_handle* h, int which, int layer, float* out, void* stream) {
  if (!h->have_hdec) return fail("kmb_attention_probs: no forward whose activations are still in the workspace");
  if (h->fp32) return fail("kmb_attention_probs: not available in the fp32 validation mode");
  const int d = h->d, B = h->bt.B;
  if (which == 0) {
    if (layer < 0 || layer >= (int)h->ea.size()) return fail("kmb_attention_probs: layer out of range");
    const EncAct& a = h->ea[layer];
    HIPCHK(kmb_attn_probs_launch(a.qkv, 3 * d, EP(a.qkv, d), 3 * d, a.lse, h->bt.attention_mask, 0, B, h->He, h->bt.S, h->bt.S, out,
                                 (hipStream_t)stream));
  } else if (which == 1) {
    if (layer < 0 || layer >= (int)h->da.size()) return fail("kmb_attention_probs: layer out of range");
    const DecAct& a = h->da[layer];
    HIPCHK(kmb_attn_probs_launch(a.qkv, 3 * d, EP(a.qkv, d), 3 * d, a.lse1, h->bt.decoder_attention_mask, 1, B, h->Hd, h->bt.T, h->bt.T,
                                 out, (hipStream_t)stream));
  } else {
    return fail("kmb_attention_probs: which must be 0 (encoder) or 1 (decoder)");
  }
  return 0;
}

int kmb_set_precision(kmb_handle* h, int fp32) {
  h->fp32 = fp32 != 0; h->have_fwd = false; h->have_hdec = false; h->gen.active = false;
  return 0;
}
int kmb_act_bytes(const kmb_handle* h) { return h->fp32 ? 4 : 2; }

// logits of the LAST forward's decoder states (outputs[1] of a training forward, src/model/model.py:397-405) without
// re-running the model: one head GEMM on the states still in the workspace
int kmb_last_logits(kmb_handle* h, float* logits_out, void* stream) {
  KCHK(check_bound(h));
  if (!h->have_hdec || h->Md <= 0) return fail("kmb_last_logits: no forward whose decoder states are still in the workspace");
  if (!logits_out) return fail("kmb_last_logits: logits_out is required");
  PrecisionScope scope(h);
  return head_logits(h, h->xd[h->cfg.decoder_layers], h->Md, logits_out, (hipStream_t)stream);
}

int kmb_reserve_head_rows(kmb_handle* h, int n) {
  if (n < 0) return fail("kmb_reserve_head_rows: negative row count");
  h->head_rows_cap = n;
  return 0;
}

int kmb_forward_pretrain(kmb_handle* h, const kmb_batch* batch, const kmb_pretrain* extra, int train, int need_grad,
                         float* logits_out, kmb_bf16* enc_out, void* stream) {
  if (!extra) return fail("kmb_forward_pretrain: extra is required");
  if ((extra->n_mrm > 0 && !h->head[0].on) || (extra->n_attr > 0 && !h->head[1].on) || (extra->n_rel > 0 && !h->head[2].on))
    return fail("kmb_forward_pretrain: rows given for a head this model was built without (num_labels / num_attributes / num_relations)");
  return forward_impl(h, batch, extra, nullptr, train, need_grad, nullptr, logits_out, enc_out, stream);
}

int kmb_forward_pretrain_ex(kmb_handle* h, const kmb_batch* batch, const kmb_pretrain* extra, const kmb_forward_opts* opts,
                            int train, int need_grad, float* logits_out, kmb_bf16* enc_out, void* stream) {
  if (!extra) return fail("kmb_forward_pretrain_ex: extra is required");
  if (opts && opts->skip_head) return fail("kmb_forward_pretrain_ex: skip_head does not apply to the pre-training forward");
  if ((extra->n_mrm > 0 && !h->head[0].on) || (extra->n_attr > 0 && !h->head[1].on) || (extra->n_rel > 0 && !h->head[2].on))
    return fail("kmb_forward_pretrain_ex: rows given for a head this model was built without (num_labels / num_attributes / num_relations)");
  return forward_impl(h, batch, extra, opts, train, need_grad, nullptr, logits_out, enc_out, stream);
}

// --------------------------------------------------------------------------------- backward
static int backward_impl(kmb_handle* h, float loss_scale, const float* loss_scale_dev, void* stream);
int kmb_backward(kmb_handle* h, float loss_scale, void* stream) { return backward_impl(h, loss_scale, nullptr, stream); }
int kmb_backward_dev(kmb_handle* h, const float* loss_scale_dev, void* stream) {
  if (!loss_scale_dev) return fail("kmb_backward_dev: loss_scale_dev is required");
  return backward_impl(h, 1.f, loss_scale_dev, stream);
}

static int backward_impl(kmb_handle* h, float loss_scale, const float* loss_scale_dev, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  KCHK(check_bound(h));
  PrecisionScope scope(h);
  if (h->small_floats > 1024) { g_small_slab = h->small_slab; g_small_floats = h->small_floats; }
  if (!h->have_fwd) return fail("kmb_backward: no forward with need_grad=1 to differentiate");
  h->have_fwd = false;
  h->have_bwd = true;
  // grouped weight gradients (wgrad_flush) for short token reductions; KMB_WGRAD_GROUP=0: every weight gradient its own
  // (split-K) launch, as for long ones
  struct GroupScope {
    kmb_handle* h;
    ~GroupScope() { h->wg_group = false; h->wg_pending.clear(); }
  } group_scope{h};
  {
    static const bool group_ok = !(getenv("KMB_WGRAD_GROUP") && getenv("KMB_WGRAD_GROUP")[0] == '0');
    static const int group_tokens = KMB_DIAG_ENV("KMB_WGRAD_GROUP_TOKENS") ? atoi(KMB_DIAG_ENV("KMB_WGRAD_GROUP_TOKENS")) : 3072;   // <= 48 samples x 64 tokens
    h->wg_group = group_ok && !h->fp32 && (h->Me > h->Md ? h->Me : h->Md) <= group_tokens;
    h->wg_pending.clear();
  }
  const kmb_batch& bt = h->bt;
  const int d = h->d, B = bt.B, S = bt.S, T = bt.T, Me = h->Me, Md = h->Md;
  const bool tr = h->fwd_train;
  const float scale = h->cfg.scale_embedding ? sqrtf((float)d) : 1.f;
  const int Le = h->cfg.encoder_layers, Ld = h->cfg.decoder_layers;
  for (auto& e : h->events)
    if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  KCHK(ensure_side(h));
  const bool side = h->side_on && h->side != nullptr;
  if (KMB_DIAG_ENV("KMB_PRINT_LAYOUT")) {   // diagnostic: workspace addresses (overlap check)
    auto pr = [&](const char* n, const void* p, size_t bytes) { fprintf(stderr, "LAYOUT %s %p %zu\n", n, p, bytes); };
    const size_t Mm = (size_t)(Me > Md ? Me : Md);
    pr("slab", h->slab, h->slab_floats * 4); pr("dhdec", h->dhdec, (size_t)Md * d * 2); pr("dyA", h->dyA, Mm * d * 2);
    pr("dyB", h->dyB, Mm * d * 2); pr("dz", h->dz, Mm * d * 2); pr("dob", h->dob, Mm * d * 2); pr("denc", h->denc, (size_t)Me * d * 2);
    pr("parts", h->parts, 0); pr("logits_c", h->logits_c, 0); pr("dlogits_c", h->dlogits_c, (size_t)Md * h->Vpad * 2);
    for (int k = 0; k < 2; ++k) {
      for (int st = 0; st < 3; ++st) { pr("bb.dz", h->bb[k].dz[st], Mm * d * 2); pr("bb.dsub", h->bb[k].dsub[st], Mm * d * 2); }
      pr("bb.du", h->bb[k].du, Mm * (size_t)(h->Fe > h->Fd ? h->Fe : h->Fd) * 2); pr("bb.dqkv", h->bb[k].dqkv, Mm * 3 * d * 2);
      pr("bb.dcq", h->bb[k].dcq, (size_t)Md * d * 2);
    }
    pr("ws_begin", h->ws, h->ws_bytes);
  }
  int ev = 0;
  // layer c may only start once the side stream has finished layer c-2 (it still reads that layer's gradient buffers)
  auto layer_begin = [&](int c) -> int {
    if (side && c >= 2) HIPCHK(hipStreamWaitEvent(s, h->layer_done[c - 2], 0));
    return 0;
  };
  // the bucket of layer c is complete when BOTH streams are past this point: record its event on the side stream
  // behind a marker of the main stream, so the main stream never waits here
  auto layer_end = [&](int c, int bucket) -> int {
    KCHK(wgrad_flush(h, s));   // the layer's parked weight gradients go out as one launch (grouped path)
    if (side) {
      hipEvent_t e = h->next_event();
      HIPCHK(hipEventRecord(e, s));
      HIPCHK(hipStreamWaitEvent(h->side, e, 0));
      HIPCHK(hipEventRecord(h->events[bucket], h->side));
      HIPCHK(hipEventRecord(h->layer_done[c], h->side));
    } else {
      HIPCHK(hipEventRecord(h->events[bucket], s));
    }
    return 0;
  };
  const bool scaled = loss_scale != 1.f || loss_scale_dev != nullptr;
  if (scaled) {
    HIPCHK(kmb_scale_bf16_launch(h->dhdec, (size_t)Md * d, loss_scale, loss_scale_dev, s));
    if (h->head_wgrad_pending) {
      // the tied matrix's head gradient is being written on the side stream: scale it there, behind the GEMM, and move
      // the completion event behind the scaling (the main stream keeps running ahead).  The device scalar was written
      // by the caller on the MAIN stream (autograd's grad_out, GradScaler's scale): the side stream must not read it
      // before the main stream got that far -- it was last ordered behind main in the middle of forward.
      if (loss_scale_dev) {
        hipEvent_t e = h->next_event();
        HIPCHK(hipEventRecord(e, s));
        HIPCHK(hipStreamWaitEvent(h->side, e, 0));
      }
      HIPCHK(kmb_scale_f32_launch(h->gf(h->shared), (size_t)h->V * d, loss_scale, loss_scale_dev, h->side));
      HIPCHK(hipEventRecord(h->head_wgrad_done, h->side));
    } else {
      HIPCHK(kmb_scale_f32_launch(h->gf(h->shared), (size_t)h->V * d, loss_scale, loss_scale_dev, s));
    }
    // the pre-training heads' parameter gradients were written by forward (head_run) at scale 1
    if (h->heads_end > h->heads_begin)
      HIPCHK(kmb_scale_f32_launch(h->gf(h->heads_begin), h->heads_end - h->heads_begin, loss_scale, loss_scale_dev, s));
  }
  const bf16_t* enc = h->xe[Le];
  const bf16_t* dy = h->dhdec;
  bf16_t* pp[2] = {h->dyA, h->dyB};
  int cur = 0, c = 0;
  bool denc_init = false;
  // ---- decoder layers
  for (int l = Ld - 1; l >= 0; --l, ++c) {
    const LayerP& L = h->dec[l];
    DecAct& a = h->da[l];
    kmb_handle::BwdBufs& bb = h->bb[c & 1];
    KCHK(layer_begin(c));
    g_trace_layer = 100 + l;
    bf16_t* t0 = pp[cur]; bf16_t* t1 = pp[cur ^ 1];
    KCHK(ffn_backward(h, L, h->Fd, a.y2, a.u, a.hh, a.z3, a.m3, a.r3, dy, t0, Md, h->drop_site(102 + 3 * l, tr), bb, s));
    {  // cross-attention block: y2 = LN(z2), z2 = y1 + drop(out_proj(attn(cq, ckv)))
      const KmbDrop dr = h->drop_site(101 + 3 * l, tr);
      bf16_t* dz = bb.dz[2];
      bf16_t* dsub = dr.thr16 ? bb.dsub[2] : dz;
      KCHK(ln_backward(h, t0, a.z2, a.m2, a.r2, L.ca.ln_g, L.ca.ln_b, dz, dr.thr16 ? dsub : nullptr,
                       KmbDrop{0u, 0u, 1.f}, dr, Md, s, L.ca.o_b, bb.parts[4]));
      KCHK(trace("ca.dz", dz, (size_t)Md * d * 2, s));
      KCHK(wgrad_side(h, lin_wgrad(dsub, d, a.o2, d, h->gf(L.ca.o_w), Md, d, d, 0.f), s));
      KmbGemm g = lin_dgrad(dsub, d, h->wb(L.ca.o_w), Md, d, d);
      g.out_bf16 = h->dob; g.ld_out_bf16 = d;
      KCHK(run_gemm(g, s));
      const int ldkv = Ld * 2 * d;
      bf16_t* dkv = h->dckv_all + (size_t)l * 2 * d;   // this layer's columns of the batched k | v gradient
      AttnIO io{a.cq, d, a.ckv, a.ckv + d, ldkv, T, S, bt.attention_mask, 0};
      KCHK(attn_backward(h, io, B, h->Hd, a.o2, a.lse2, h->dob, bb.dcq, d, dkv, dkv + d, ldkv, bb.parts[5],
                         bb.parts[5] + d, bb.parts[5] + 2 * d, 3 * d, s));
      // bias gradients from the attention kernel's per-batch-item column sums [B][q | k | v]: q's bias lives in the layer,
      // the k | v biases of all layers together (kmb_handle::xkv_b)
      HIPCHK(kmb_reduce_parts2_launch(bb.parts[5], B, 3 * d, h->gf(L.ca.qkv_b), d, h->gf(L.ca_kv_b), 2 * d, reducer_stream(h, s)));
      KCHK(trace("ca.dob", h->dob, (size_t)Md * d * 2, s));
      KCHK(trace("ca.dcq", bb.dcq, (size_t)Md * d * 2, s));
      KCHK(wgrad_side(h, lin_wgrad(bb.dcq, d, a.y1, d, h->gf(L.ca.qkv_w), Md, d, d, 0.f), s));
      g = lin_dgrad(bb.dcq, d, h->wb(L.ca.qkv_w), Md, d, d);
      g.residual = dz; g.ld_res = d; g.out_bf16 = t1; g.ld_out_bf16 = d;
      KCHK(run_gemm(g, s));
      KCHK(trace("ca.t1", t1, (size_t)Md * d * 2, s));
    }
    KCHK(self_attn_backward(h, L.sa, h->Hd, h->xd[l], a.qkv, a.o1, a.lse1, a.z1, a.m1, a.r1, t1, t0, B, T,
                            bt.decoder_attention_mask, 1, h->drop_site(100 + 3 * l, tr), bb, s));
    dy = t0;  // t0 now holds d(loss)/d(xd[l]); keep it as the input of the next iteration
    cur ^= 1;  // next iteration writes its first result into the other buffer
    KCHK(layer_end(c, ev++));
  }
  // ---- cross-attention keys | values of ALL decoder layers: one data gradient (the encoder output's gradient, reduction
  // over Ld * 2d) and one weight gradient
  if (Ld > 0) {
    const int ldkv = Ld * 2 * d;
    KCHK(trace("ca.dckv_all", h->dckv_all, (size_t)Me * ldkv * 2, s));
    KCHK(wgrad_side(h, lin_wgrad(h->dckv_all, ldkv, enc, d, h->gf(h->xkv_w), Me, ldkv, d, 0.f), s));
    KmbGemm g = lin_dgrad(h->dckv_all, ldkv, h->wb(h->xkv_w), Me, ldkv, d);
    g.out_bf16 = h->denc; g.ld_out_bf16 = d;
    KCHK(run_gemm(g, s));
    KCHK(trace("ca.denc", h->denc, (size_t)Me * d * 2, s));
    denc_init = true;
  }
  // ---- decoder embedding: xd[0] = drop(LN(zd0))  (main stream only)
  KCHK(ln_backward(h, dy, h->zd0, h->md0, h->rd0, h->dec_lne_g, h->dec_lne_b, h->dz, nullptr, h->drop_site(2, tr),
                   KmbDrop{0u, 0u, 1.f}, Md, s));
  if (h->head_wgrad_pending) {
    HIPCHK(hipStreamWaitEvent(s, h->head_wgrad_done, 0));
    h->head_wgrad_pending = false;
  }
  HIPCHK(kmb_embed_bwd_launch(h->dz, bt.decoder_input_ids, nullptr, scale, h->gf(h->shared), nullptr,
                              h->cfg.pad_token_id, Md, d, s));
  HIPCHK(kmb_pos_bwd_launch(h->dz, B, T, d, h->gf(h->dec_pos), h->cfg.extra_pos_embeddings, h->Prows, s));
  // this bucket also holds the cross-attention k | v weights and biases of every decoder layer, whose gradients come from
  // the side stream (the batched weight gradient above, the per-layer bias reducers): complete when BOTH streams are here
  KCHK(wgrad_flush(h, s));
  if (side) {
    hipEvent_t e = h->next_event();
    HIPCHK(hipEventRecord(e, s));
    HIPCHK(hipStreamWaitEvent(h->side, e, 0));
    HIPCHK(hipEventRecord(h->events[ev++], h->side));
  } else {
    HIPCHK(hipEventRecord(h->events[ev++], s));
  }
  // ---- encoder layers
  if (!denc_init) HIPCHK(hipMemsetAsync(h->denc, 0, (size_t)Me * d * sizeof(bf16_t), s));
  if (h->enc_given) {
    // The forward started from the caller's encoder states: the gradient stops at them (h->denc, kmb_encoder_states_grad).
    // The encoder's parameters took no part: their buckets are zero, complete here (the tied matrix keeps the decoder side's
    // contributions: the head's weight gradient and the decoder embedding's scatter-add above).
    for (int l = Le - 1; l >= 0; --l, ++ev) {
      HIPCHK(hipMemsetAsync(h->gf(h->buckets[ev].off), 0, h->buckets[ev].count * sizeof(float), s));
      HIPCHK(hipEventRecord(h->events[ev], s));
    }
    HIPCHK(hipMemsetAsync(h->gf(h->buckets[ev].off), 0, h->buckets[ev].count * sizeof(float), s));
    if (side) {
      hipEvent_t e = h->next_event();
      HIPCHK(hipEventRecord(e, h->side));
      HIPCHK(hipStreamWaitEvent(s, e, 0));
    }
    HIPCHK(hipEventRecord(h->events[ev++], s));
    HIPCHK(hipEventRecord(h->events[ev++], s));
    return 0;
  }
  dy = h->denc;
  cur = 0;
  for (int l = Le - 1; l >= 0; --l, ++c) {
    const LayerP& L = h->enc[l];
    EncAct& a = h->ea[l];
    kmb_handle::BwdBufs& bb = h->bb[c & 1];
    KCHK(layer_begin(c));
    g_trace_layer = l;
    bf16_t* t0 = pp[cur]; bf16_t* t1 = pp[cur ^ 1];
    KCHK(ffn_backward(h, L, h->Fe, a.y1, a.u, a.hh, a.z2, a.m2, a.r2, dy, t0, Me, h->drop_site(11 + 2 * l, tr), bb, s));
    KCHK(self_attn_backward(h, L.sa, h->He, h->xe[l], a.qkv, a.o, a.lse, a.z1, a.m1, a.r1, t0, t1, B, S,
                            bt.attention_mask, 0, h->drop_site(10 + 2 * l, tr), bb, s));
    dy = t1;  // t0 / t1 keep their roles: the next ffn_backward reads t1 and writes t0
    KCHK(layer_end(c, ev++));
  }
  // ---- encoder embedding (+ image projection, src/model/modules.py:24-41)
  KCHK(ln_backward(h, dy, h->ze0, h->me0, h->re0, h->enc_lne_g, h->enc_lne_b, h->dz, nullptr, h->drop_site(1, tr),
                   KmbDrop{0u, 0u, 1.f}, Me, s));
  HIPCHK(kmb_embed_bwd_launch(h->dz, bt.input_ids, h->img_src, scale, h->gf(h->shared), h->dimg, h->cfg.pad_token_id,
                              Me, d, s));
  HIPCHK(kmb_pos_bwd_launch(h->dz, B, S, d, h->gf(h->enc_pos), h->cfg.extra_pos_embeddings, h->Prows, s));
  if (h->Ntot > 0) {
    KCHK(bias_grad(h, h->dimg, d, h->Ntot, d, h->gf(h->img_b), s));
    KCHK(wgrad_side(h, lin_wgrad(h->dimg, d, h->xf, h->Fpad, h->gf(h->img_w), h->Ntot, d, h->Fin, 0.f), s));  // shares the slab
  } else {
    HIPCHK(hipMemsetAsync(h->gf(h->img_w), 0, ((size_t)d * h->Fin) * sizeof(float), s));
    HIPCHK(hipMemsetAsync(h->gf(h->img_b), 0, (size_t)d * sizeof(float), s));
  }
  KCHK(wgrad_flush(h, s));
  if (side) {  // everything the optimizer reads must be ordered behind the side stream's last weight gradient
    hipEvent_t e = h->next_event();
    HIPCHK(hipEventRecord(e, h->side));
    HIPCHK(hipStreamWaitEvent(s, e, 0));
  }
  HIPCHK(hipEventRecord(h->events[ev++], s));
  HIPCHK(hipEventRecord(h->events[ev++], s));  // tied matrix: complete once the encoder-side scatter-add is in
  return 0;
}

// diagnostic: start (on = 1) / stop recording buffer checksums in backward; dump prints "index layer name checksum"
int kmb_debug_trace(int on) {
  g_trace_on = on != 0;
  if (on) g_trace.clear();
  return 0;
}
int kmb_debug_trace_dump(const char* path) {
  HIPCHK(hipDeviceSynchronize());
  std::vector<unsigned long long> hst(g_trace.size());
  if (!hst.empty()) HIPCHK(hipMemcpy(hst.data(), g_trace_dev, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  FILE* f = fopen(path, "w");
  if (!f) return fail("kmb_debug_trace_dump: cannot open %s", path);
  for (size_t i = 0; i < hst.size(); ++i) fprintf(f, "%zu %d %s %016llx\n", i, g_trace[i].layer, g_trace[i].name, hst[i]);
  fclose(f);
  return 0;
}

// 0: weight-gradient GEMMs stay on the caller's stream (serial backward; used while timing single kernels)
int kmb_set_side_stream(kmb_handle* h, int enable) {
  h->side_on = enable != 0;
  return 0;
}

int kmb_profile_gemm(int enable) {
  g_prof.on = enable != 0;
  if (enable) { g_prof.used = 0; g_prof.recs.clear(); }
  return 0;
}

// variant index = a_kc*2 + b_kc (3: forward, 2: dgrad, 0: wgrad).  Synchronises the events.
int kmb_profile_read(int variant, int64_t* launches, double* total_ms, double* total_flops) {
  int64_t n = 0; double ms = 0, fl = 0;
  for (size_t i = 0; i < g_prof.recs.size(); ++i) {
    if (g_prof.recs[i].variant != variant) continue;
    float t = 0.f;
    const size_t pr = g_prof.recs[i].pair;
    HIPCHK(hipEventSynchronize(g_prof.ev[2 * pr + 1]));
    HIPCHK(hipEventElapsedTime(&t, g_prof.ev[2 * pr], g_prof.ev[2 * pr + 1]));
    ms += t * g_prof.recs[i].share; fl += g_prof.recs[i].flops; n += g_prof.recs[i].launch;
  }
  *launches = n; *total_ms = ms; *total_flops = fl;
  return 0;
}

// one line per GEMM problem of the profiled calls: variant M N K split act microseconds residual group -- `group` = n for a
// problem that went out as one of the n of a grouped weight-gradient launch (its microseconds are its FLOP share of that
// launch), 1 for a launch of its own
int kmb_profile_dump(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return fail("kmb_profile_dump: cannot open %s", path);
  for (size_t i = 0; i < g_prof.recs.size(); ++i) {
    float t = 0.f;
    const auto& r = g_prof.recs[i];
    HIPCHK(hipEventSynchronize(g_prof.ev[2 * r.pair + 1]));
    HIPCHK(hipEventElapsedTime(&t, g_prof.ev[2 * r.pair], g_prof.ev[2 * r.pair + 1]));
    fprintf(f, "%d %d %d %d %d %d %.3f %d %d\n", r.variant, r.M, r.N, r.K, r.split, r.act, t * r.share * 1e3, r.res, r.group_n);   // res: the epilogue reads a residual operand
  }
  fclose(f);
  return 0;
}

int kmb_adamw_step(kmb_handle* h, const KmbAdamW* hp, int64_t offset, int64_t count, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h->P || !h->G || !h->M1 || !h->M2) return fail("kmb_adamw_step: arenas are not bound");
  if (offset < 0 || count < 0 || (size_t)(offset + count) > h->arena) return fail("kmb_adamw_step: range out of the arena");
  if (offset & 7) return fail("kmb_adamw_step: offset must be a multiple of 8 elements");
  HIPCHK(kmb_adamw_launch(h->P + offset, h->G + offset, h->M1 + offset, h->M2 + offset, h->PB + offset, (size_t)count, *hp, s));
  h->mirror_version++;
  // the padded image-weight mirror lives outside the flat mirror
  const size_t iw0 = h->img_w, iw1 = h->img_w + (size_t)h->d * h->Fin;
  if ((size_t)offset < iw1 && (size_t)(offset + count) > iw0)
    HIPCHK(kmb_cast_rows_launch(h->pf(h->img_w), h->Fin, h->imgw_pad, h->Fpad, h->d, h->Fin, s));
  return 0;
}

}  // extern "C"


// ================================================================================= native data parallelism (RCCL)
namespace {

#define NCCLCHK(expr)                                                                                          \
  do {                                                                                                         \
    ncclResult_t r_ = (expr);                                                                                  \
    if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

struct Piece { int bucket; size_t off, cnt; };

// the collectives of one gradient exchange: every bucket (backward completion order) cut into pieces of at most `cap`
// elements whose boundaries stay 64-element aligned (the fused optimizer's and the shards' alignment)
std::vector<Piece> comm_pieces(const kmb_handle* h, int64_t max_piece_elems) {
  const size_t cap = max_piece_elems > 0 ? (size_t)max_piece_elems : (size_t)16 << 20;
  std::vector<Piece> out;
  for (size_t i = 0; i < h->buckets.size(); ++i) {
    const size_t off = h->buckets[i].off, cnt = h->buckets[i].count;
    if (cnt == 0) continue;
    if (cnt > cap) {
      const size_t n = (cnt + cap - 1) / cap;
      const size_t step = align_up((cnt + n - 1) / n, 64);
      for (size_t s0 = 0; s0 < cnt; s0 += step) out.push_back({(int)i, off + s0, std::min(step, cnt - s0)});
    } else {
      out.push_back({(int)i, off, cnt});
    }
  }
  return out;
}

// piece i of the exchange as rank `rank` of `world` sees it: a pure function of the arena layout (no communicator, no device).
// Every collective of kmb_allreduce_grads / kmb_comm_gather_moments takes its offsets from here, so the world > 1 arithmetic
// is testable on a CPU (tests/test_comm_plan_cpu.py).
int comm_plan_piece(const kmb_handle* h, const Piece& pc, int world, int rank, kmb_comm_piece* out) {
  const size_t W = (size_t)world;
  out->bucket = pc.bucket; out->offset = (int64_t)pc.off; out->count = (int64_t)pc.cnt;
  const size_t shard = pc.cnt / W;   // pieces are multiples of 64 elements and world divides 8: shards stay 8-element aligned
  out->shard = (shard * W == pc.cnt && (shard & 7) == 0) ? (int64_t)shard : 0;   // 0: this piece has no aligned shards (algo 1 refuses it)
  out->mine = (int64_t)(pc.off + (size_t)rank * shard);
  const size_t iw0 = h->img_w, iw1 = h->img_w + (size_t)h->d * h->Fin;
  out->repad_piece = (pc.off < iw1 && pc.off + pc.cnt > iw0) ? 1 : 0;
  out->repad_shard = ((size_t)out->mine < iw1 && (size_t)out->mine + shard > iw0) ? 1 : 0;
  return 0;
}

int comm_ready(const kmb_handle* h, const char* who) {
  if (!h->comm || !h->comm_stream) return fail("%s: no communicator (call kmb_comm_init first)", who);
  if (!h->P || !h->G) return fail("%s: arenas are not bound", who);
  return 0;
}

// the optimizer update of arena range [off, off + cnt) on stream s (what kmb_adamw_step does, shared with it)
int adamw_range(kmb_handle* h, const KmbAdamW& hp, size_t off, size_t cnt, hipStream_t s) {
  if (!h->M1 || !h->M2) return fail("fused optimizer step: the moment arenas are not bound");
  HIPCHK(kmb_adamw_launch(h->P + off, h->G + off, h->M1 + off, h->M2 + off, h->PB + off, cnt, hp, s));
  h->mirror_version++;
  const size_t iw0 = h->img_w, iw1 = h->img_w + (size_t)h->d * h->Fin;
  if (off < iw1 && off + cnt > iw0)
    HIPCHK(kmb_cast_rows_launch(h->pf(h->img_w), h->Fin, h->imgw_pad, h->Fpad, h->d, h->Fin, s));
  return 0;
}

}  // namespace

extern "C" {

int kmb_comm_unique_id(void* id_host) {
  static_assert(sizeof(ncclUniqueId) == KMB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id_host) return fail("kmb_comm_unique_id: id_host is required");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(id_host, &id, sizeof(id));
  return 0;
}

int kmb_comm_init(kmb_handle* h, int rank, int world, const void* id_host) {
  if (!id_host || world < 1 || rank < 0 || rank >= world) return fail("kmb_comm_init: bad rank / world / id");
  if (h->comm) return fail("kmb_comm_init: this handle already has a communicator");
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  NCCLCHK(ncclCommInitRank(&h->comm, world, id, rank));
  h->comm_rank = rank; h->comm_world = world;
  HIPCHK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&h->comm_ev, hipEventDisableTiming));
  return 0;
}

int kmb_comm_destroy(kmb_handle* h) {
  if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
  if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
  if (h->comm_ev) { (void)hipEventDestroy(h->comm_ev); h->comm_ev = nullptr; }
  h->comm_world = 0; h->comm_rank = 0;
  //  without a communicator nothing can gather the moment shards any more: the flag must not outlive it, or a later plain
  // AdamW.step() / state_dict() would ask for a collective on a destroyed communicator.  The Python wrapper gathers BEFORE it destroys
  // (kmbart/engine.py::comm_destroy); a caller of the C-ABI that destroys with sharded moments keeps this rank's shard only.
  h->moments_sharded = false;
  return 0;
}

int kmb_comm_info(const kmb_handle* h, int32_t* rank, int32_t* world) {
  if (rank) *rank = h->comm_rank;
  if (world) *world = h->comm ? h->comm_world : 0;
  return 0;
}

int kmb_comm_broadcast_params(kmb_handle* h, int root, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  KCHK(comm_ready(h, "kmb_comm_broadcast_params"));
  if (root < 0 || root >= h->comm_world) return fail("kmb_comm_broadcast_params: root out of range");
  NCCLCHK(ncclBroadcast(h->P, h->P, h->arena, ncclFloat, root, h->comm, s));
  NCCLCHK(ncclBroadcast(h->flb, h->flb, (size_t)h->V, ncclFloat, root, h->comm, s));
  return kmb_sync_params(h, stream);
}

int64_t kmb_comm_pieces(const kmb_handle* h, int64_t max_piece_elems) { return (int64_t)comm_pieces(h, max_piece_elems).size(); }

int kmb_comm_plan(const kmb_handle* h, int world, int rank, int64_t max_piece_elems, int64_t i, kmb_comm_piece* out) {
  if (!h || !out) return fail("kmb_comm_plan: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("kmb_comm_plan: bad rank %d / world %d", rank, world);
  const std::vector<Piece> pcs = comm_pieces(h, max_piece_elems);
  if (i < 0 || i >= (int64_t)pcs.size()) return fail("kmb_comm_plan: piece %lld out of range (%zu pieces)", (long long)i, pcs.size());
  return comm_plan_piece(h, pcs[(size_t)i], world, rank, out);
}
int kmb_comm_moments_sharded(const kmb_handle* h) { return h && h->moments_sharded ? 1 : 0; }

int kmb_allreduce_grads(kmb_handle* h, const kmb_allreduce_opts* opts, void* compute_stream) {
  KCHK(comm_ready(h, "kmb_allreduce_grads"));
  static const kmb_allreduce_opts dflt{0, 0, 0, nullptr};
  const kmb_allreduce_opts& o = opts ? *opts : dflt;
  const int W = h->comm_world, rank = h->comm_rank;
  if (o.algo != 0 && o.algo != 1) return fail("kmb_allreduce_grads: algo must be 0 (all-reduce) or 1 (reduce-scatter + all-gather)");
  if (o.algo == 1 && (8 % W) != 0) return fail("kmb_allreduce_grads: algo 1 needs a world size that divides 8 (got %d)", W);
  if (o.algo == 1) h->comm_piece_cap = o.max_piece_elems;
  hipStream_t cs = h->comm_stream;
  if (o.after_compute) {
    HIPCHK(hipEventRecord(h->comm_ev, (hipStream_t)compute_stream));
    HIPCHK(hipStreamWaitEvent(cs, h->comm_ev, 0));
  }
  int waited = -1;
  for (const Piece& pc : comm_pieces(h, o.max_piece_elems)) {
    if (pc.bucket != waited) {   // the bucket's gradients are complete (both compute streams are past it: kmb_backward)
      if (!h->events[pc.bucket]) return fail("kmb_allreduce_grads: bucket %d has no completion event (run kmb_backward first)", pc.bucket);
      HIPCHK(hipStreamWaitEvent(cs, h->events[pc.bucket], 0));
      waited = pc.bucket;
    }
    float* g = h->G + pc.off;
    if (o.algo == 0) {
      NCCLCHK(ncclAllReduce(g, g, pc.cnt, ncclFloat, ncclAvg, h->comm, cs));
      if (o.adamw) KCHK(adamw_range(h, *o.adamw, pc.off, pc.cnt, cs));
      continue;
    }
    // reduce-scatter: rank r ends up with the mean of shard r (in place: recvbuff = sendbuff + r * shard)
    kmb_comm_piece pl;
    KCHK(comm_plan_piece(h, pc, W, rank, &pl));
    if (pl.shard == 0) return fail("kmb_allreduce_grads: piece of %zu elements does not split into %d aligned shards", pc.cnt, W);
    const size_t shard = (size_t)pl.shard, mine = (size_t)pl.mine;
    NCCLCHK(ncclReduceScatter(g, h->G + mine, shard, ncclFloat, ncclAvg, h->comm, cs));
    if (o.adamw) {
      KCHK(adamw_range(h, *o.adamw, mine, shard, cs));                                              // 30 B / parameter / W
      NCCLCHK(ncclAllGather(h->P + mine, h->P + pc.off, shard, ncclFloat, h->comm, cs));           // updated fp32 masters
      if (W > 1) {   // the other ranks' shards of the bf16 mirror (this rank's was written by the optimizer kernel)
        HIPCHK(kmb_cast_f32_bf16_launch(h->P + pc.off, h->PB + pc.off, pc.cnt, cs));
        h->mirror_version++;
        if (pl.repad_piece)
          HIPCHK(kmb_cast_rows_launch(h->pf(h->img_w), h->Fin, h->imgw_pad, h->Fpad, h->d, h->Fin, cs));
        h->moments_sharded = true;   // exp_avg / exp_avg_sq are current on the owning rank's shard only
      }
    } else {
      NCCLCHK(ncclAllGather(h->G + mine, g, shard, ncclFloat, h->comm, cs));
    }
  }
  return 0;
}

int kmb_comm_wait(kmb_handle* h, void* compute_stream) {
  if (!h->comm_stream) return fail("kmb_comm_wait: no communicator");
  HIPCHK(hipEventRecord(h->comm_ev, h->comm_stream));
  HIPCHK(hipStreamWaitEvent((hipStream_t)compute_stream, h->comm_ev, 0));
  return 0;
}

int kmb_comm_gather_moments(kmb_handle* h, void* compute_stream) {
  KCHK(comm_ready(h, "kmb_comm_gather_moments"));
  if (!h->M1 || !h->M2) return fail("kmb_comm_gather_moments: the moment arenas are not bound");
  const int W = h->comm_world, rank = h->comm_rank;
  if ((8 % W) != 0) return fail("kmb_comm_gather_moments: world size must divide 8");
  hipStream_t s = (hipStream_t)compute_stream;
  for (const Piece& pc : comm_pieces(h, h->comm_piece_cap)) {
    kmb_comm_piece pl;
    KCHK(comm_plan_piece(h, pc, W, rank, &pl));
    if (pl.shard == 0) return fail("kmb_comm_gather_moments: piece of %zu elements does not split into %d aligned shards", pc.cnt, W);
    NCCLCHK(ncclAllGather(h->M1 + pl.mine, h->M1 + pc.off, (size_t)pl.shard, ncclFloat, h->comm, s));
    NCCLCHK(ncclAllGather(h->M2 + pl.mine, h->M2 + pc.off, (size_t)pl.shard, ncclFloat, h->comm, s));
  }
  h->moments_sharded = false;
  return 0;
}

}  // extern "C"

// ================================================================================= generation
namespace {

struct GenLayout {
  int32_t* status; bf16_t* xf; float* img_emb; int32_t* img_src; bf16_t* xe[2]; EncAct ea;
  std::vector<bf16_t*> ckv, kc[2], vc[2];
  int32_t* kv_row; bf16_t *x0, *x1, *qkv, *o, *z, *y, *cq, *u, *hh; float *mean, *rstd;
  float* slab;   // split-K partial sums of the residual projections of a decode step
  std::vector<bf16_t*> wp;   // per layer: self q|k|v, self out, cross q, cross out, fc1, fc2 in fragment order (decode.hip)
  uint32_t* bars;            // group-barrier counters of the resident decoder-layers kernel (decode.hip)
  int32_t* hist[2];          // history index of the self-attention caches [R, Tmax], ping-pong over beam reorders
  float* head_stats;         // the all-rows vocabulary projection's per-block (maximum, sum-exp) pairs (kmb_gen_beam_step)
};
constexpr int GEN_MAX_SPLIT = 12;

// the fused decode blocks need d_model = 768 (one 768-deep weight block per attention projection, 64-wide heads) and an
// FFN width of 768 .. 3072 in steps of 768
bool gen_fused_eligible(const kmb_handle* h) {
  return h->d == 768 && h->Hd * 64 == h->d && (h->Fd % 768) == 0 && h->Fd <= 3072 && (h->Fd % 64) == 0;
}

size_t layout_gen(const kmb_handle* h, char* base, size_t cap, int B, int S, int nb, int Tmax, int Ntot, GenLayout* out) {
  const int d = h->d, Fe = h->Fe, Fd = h->Fd, Ld = h->cfg.decoder_layers;
  const size_t Me = (size_t)B * S, R = (size_t)B * nb;
  Bump bp(base, cap);
  GenLayout g;
  g.status = bp.take<int32_t>(4);
  g.xf = bp.act((size_t)(Ntot > 0 ? Ntot : 1) * h->Fpad);
  g.img_emb = bp.take<float>((size_t)(Ntot > 0 ? Ntot : 1) * d);
  g.img_src = bp.take<int32_t>(Me);
  g.xe[0] = bp.act(Me * d); g.xe[1] = bp.act(Me * d);
  EncAct& a = g.ea;
  a.qkv = bp.act(Me * 3 * d); a.o = bp.act(Me * d); a.z1 = bp.act(Me * d);
  a.y1 = bp.act(Me * d); a.u = bp.act(Me * Fe); a.hh = bp.act(Me * Fe);
  a.z2 = bp.act(Me * d); a.lse = bp.take<float>((size_t)B * h->He * S);
  a.m1 = bp.take<float>(Me); a.r1 = bp.take<float>(Me); a.m2 = bp.take<float>(Me); a.r2 = bp.take<float>(Me);
  g.ckv.resize(Ld);
  for (int i = 0; i < 2; ++i) { g.kc[i].resize(Ld); g.vc[i].resize(Ld); }
  bf16_t* ckv_all = bp.act(Me * (size_t)Ld * 2 * d);   // [Me, Ld * 2d]: layer l's cross-attention k | v are columns [l * 2d, (l + 1) * 2d)
  for (int l = 0; l < Ld; ++l) {
    g.ckv[l] = ckv_all + (size_t)l * 2 * d;
    for (int i = 0; i < 2; ++i) { g.kc[i][l] = bp.act(R * Tmax * d); g.vc[i][l] = bp.act(R * Tmax * d); }
  }
  g.kv_row = bp.take<int32_t>(2 * R);   // two copies: a reorder of independent rows (num_beams == 1) gathers it into the other one
  g.x0 = bp.act(R * d); g.x1 = bp.act(R * d); g.qkv = bp.act(R * 3 * d);
  g.o = bp.act(R * d); g.z = bp.act(R * d); g.y = bp.act(R * d); g.cq = bp.act(R * d);
  g.u = bp.act(R * Fd); g.hh = bp.act(R * Fd);
  g.mean = bp.take<float>(R); g.rstd = bp.take<float>(R);
  g.slab = bp.take<float>((size_t)GEN_MAX_SPLIT * R * d);
  if (gen_fused_eligible(h)) {
    const size_t dd = (size_t)d * d, fd = (size_t)Fd * d;
    const size_t sizes[6] = {3 * dd, dd, dd, dd, fd, fd};
    for (int l = 0; l < Ld; ++l)
      for (int i = 0; i < 6; ++i) g.wp.push_back(bp.act(sizes[i]));
  }
#ifdef KMB_WITH_RESIDENT_DECODE   // experiment build only (tools/experiments/decode_resident.hip): the group-barrier counters
  g.bars = bp.take<uint32_t>(kmb_decode_layers_bar_words((int)R, Ld > 0 ? Ld : 1) + 64);
#endif
  g.hist[0] = bp.take<int32_t>(R * Tmax); g.hist[1] = bp.take<int32_t>(R * Tmax);
  g.head_stats = bp.take<float>(kmb_gemm_allrows_stats_floats(h->V));
  if (out) *out = g;
  return bp.used();
}

}  // namespace

extern "C" {

int64_t kmb_gen_workspace_bytes(const kmb_handle* h, int B, int S, int num_beams, int max_length, int n_features) {
  return (int64_t)layout_gen(h, nullptr, 0, B, S, num_beams, max_length, n_features, nullptr);
}

int kmb_gen_begin(kmb_handle* h, const kmb_batch* batch, int num_beams, int max_length, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  KCHK(check_bound(h));
  if (!batch || !batch->input_ids || !batch->feat_offsets) return fail("kmb_gen_begin: input_ids and feat_offsets are required");
  if (num_beams < 1 || max_length < 2) return fail("kmb_gen_begin: bad num_beams / max_length");
  if (max_length > h->cfg.max_position_embeddings) return fail("kmb_gen_begin: max_length exceeds max_position_embeddings");
  const kmb_batch& bt = *batch;
  const int d = h->d, B = bt.B, S = bt.S, Me = B * S;
  GenLayout g;
  const size_t need = layout_gen(h, h->ws, h->ws_bytes, B, S, num_beams, max_length, bt.n_features, &g);
  if (need > h->ws_bytes) return fail("kmb_gen_begin: workspace too small (%zu > %zu bytes)", need, h->ws_bytes);
  if (h->fp32) return fail("kmb_gen_begin: generation is not available in the fp32 validation mode");
  h->have_fwd = false; h->have_hdec = false;
  // point the encoder sub-graph at the (layer-shared) generation buffers
  const int Le = h->cfg.encoder_layers, Ld = h->cfg.decoder_layers;
  h->status = g.status; h->xf = g.xf; h->img_emb = g.img_emb; h->img_src = g.img_src;
  h->ze0 = nullptr; h->me0 = nullptr; h->re0 = nullptr;
  h->xe.assign(Le + 1, nullptr);
  for (int l = 0; l <= Le; ++l) h->xe[l] = g.xe[l & 1];
  h->ea.assign(Le, g.ea);
  HIPCHK(hipMemsetAsync(h->status, 0, 16, s));
  KCHK(encoder_forward(h, bt, false, s));
  const bf16_t* enc = h->xe[Le];
  auto& G = h->gen;
  G.active = true; G.B = B; G.S = S; G.nb = num_beams; G.R = B * num_beams; G.Tmax = max_length; G.bt = bt; G.cur = 0;
  G.ckv = g.ckv; G.kc[0] = g.kc[0]; G.kc[1] = g.kc[1]; G.vc[0] = g.vc[0]; G.vc[1] = g.vc[1];
  G.kv_row = g.kv_row; G.kv_row_base = g.kv_row; G.x0 = g.x0; G.x1 = g.x1; G.qkv = g.qkv; G.o = g.o; G.z = g.z; G.y = g.y; G.cq = g.cq;
  G.u = g.u; G.hh = g.hh; G.mean = g.mean; G.rstd = g.rstd; G.slab = g.slab; G.wp = g.wp;
  G.last_x = nullptr; G.last_z = nullptr; G.last_g = nullptr; G.last_b = nullptr;
  G.bars = g.bars;
  G.hist[0] = g.hist[0]; G.hist[1] = g.hist[1]; G.hcur = 0;
  G.head_stats = g.head_stats; G.head_stats_blocks = 0; G.head_stats_for = nullptr;
  G.x0_step = -1; G.x0_tokens = nullptr;
  { const char* he = getenv("KMB_GEN_HIST"); G.use_hist = !(he && he[0] == '0'); }
  // cross-attention K|V of every decoder layer, computed once per batch item (not per beam), all layers in ONE GEMM
  if (Ld > 0) {
    KmbGemm gm = lin_fwd(enc, d, h->wb(h->xkv_w), h->pf(h->xkv_b), Me, Ld * 2 * d, d);
    gm.out_bf16 = G.ckv[0]; gm.ld_out_bf16 = Ld * 2 * d;
    KCHK(run_gemm(gm, s));
  }
  // (skipped when the copies of the last kmb_gen_begin are still there: same place in the workspace, the bf16 mirror unchanged since,
  //  no training forward in between -- that one re-uses the workspace and clears packed_at)
  if (!G.wp.empty() && !(G.packed_at == G.wp[0] && G.packed_version == h->mirror_version)) {   // fragment-order copies of the decoder weights for the fused decode blocks, one launch per 48
    std::vector<const bf16_t*> src; std::vector<bf16_t*> dst; std::vector<int> ld, nn, kk;
    for (int l = 0; l < Ld; ++l) {
      const LayerP& L = h->dec[l];
      const size_t offs[6] = {L.sa.qkv_w, L.sa.o_w, L.ca.qkv_w, L.ca.o_w, L.fc1_w, L.fc2_w};
      const int N6[6] = {3 * d, d, d, d, h->Fd, d}, K6[6] = {d, d, d, d, d, h->Fd};
      for (int i = 0; i < 6; ++i) {
        src.push_back(h->wb(offs[i])); dst.push_back(G.wp[(size_t)l * 6 + i]); ld.push_back(K6[i]); nn.push_back(N6[i]); kk.push_back(K6[i]);
      }
    }
    for (size_t i0 = 0; i0 < src.size(); i0 += 48) {
      const int n = (int)std::min<size_t>(48, src.size() - i0);
      HIPCHK(kmb_decode_pack_launch(src.data() + i0, ld.data() + i0, nn.data() + i0, kk.data() + i0, dst.data() + i0, n, s));
    }
    G.packed_at = G.wp[0]; G.packed_version = h->mirror_version;
  }
  // beam row -> batch item, written on the device: a host table needed a copy and a stream synchronisation here, and the
  // host then sat out the encoder (1.1 ms at batch 64) instead of queueing the first decode steps behind it
  HIPCHK(kmb_iota_div_launch(G.kv_row, G.R, num_beams, s));
  return 0;
}

int kmb_gen_encoder_states(kmb_handle* h, kmb_bf16* enc_out, void* stream) {
  auto& G = h->gen;
  if (!G.active) return fail("kmb_gen_encoder_states: call kmb_gen_begin first");
  if (!enc_out) return fail("kmb_gen_encoder_states: enc_out is required");
  const bf16_t* enc = h->xe[h->cfg.encoder_layers];
  HIPCHK(hipMemcpyAsync(enc_out, enc, (size_t)G.B * G.S * h->d * sizeof(bf16_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int kmb_gen_step(kmb_handle* h, const int64_t* tokens, int step, float* logits_out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  auto& G = h->gen;
  if (!G.active) return fail("kmb_gen_step: call kmb_gen_begin first");
  if (step < 0 || step >= G.Tmax) return fail("kmb_gen_step: step %d outside the cache (Tmax=%d)", step, G.Tmax);
  const int d = h->d, R = G.R, F = h->Fd;
  const float eps = h->cfg.layer_norm_eps;
  const float scale = h->cfg.scale_embedding ? sqrtf((float)d) : 1.f;
  const KmbDrop nodrop{0u, 0u, 1.f};
  // BartDecoder with use_cache: only the last token, learned position (len-1) + 2
  // (skipped when the beam step that chose these tokens has embedded them already: kmb_gen_beam_step, same row code)
  if (!(G.x0_step == step && G.x0_tokens == tokens))
    HIPCHK(kmb_embed_ln_fwd_launch(tokens, nullptr, h->pf(h->shared), nullptr, h->pf(h->dec_pos),
                                   h->cfg.extra_pos_embeddings + step, 1, scale, h->pf(h->dec_lne_g),
                                   h->pf(h->dec_lne_b), nullptr, G.x0, nullptr, nullptr, R, d, eps, nodrop, s));
  G.x0_step = -1; G.x0_tokens = nullptr;
  bf16_t* x = G.x0; bf16_t* xn = G.x1;
  // residual projection + LayerNorm (BartDecoderLayer: x = LN(residual + dropout(proj(x)))).  With R = batch x beams rows
  // the projection has 18 output tiles of 128 x 128 and a serial K loop: split K over workgroups and let ONE kernel sum
  // the slabs, add bias and residual and normalise (no GEMM epilogue, no separate LayerNorm launch).
  auto proj_ln = [&](const bf16_t* in, int K, size_t w_off, size_t b_off, const bf16_t* res, size_t g_off, size_t be_off,
                     bf16_t* out) -> int {
    const int nt = K / 64;
    static const int s_small = KMB_DIAG_ENV("KMB_GEN_SPLIT_SMALL") ? atoi(KMB_DIAG_ENV("KMB_GEN_SPLIT_SMALL")) : 3;   // tuning knobs
    static const int s_large = KMB_DIAG_ENV("KMB_GEN_SPLIT_LARGE") ? atoi(KMB_DIAG_ENV("KMB_GEN_SPLIT_LARGE")) : 6;
    int S = K >= 2048 ? s_large : s_small;
    if (S > nt / 2) S = nt / 2;
    if (S > GEN_MAX_SPLIT) S = GEN_MAX_SPLIT;
    if (S > 1 && (K % 64) == 0 && (d & 7) == 0 && d <= 1024) {
      KmbGemm g = lin_fwd(in, K, h->wb(w_off), nullptr, R, d, K);
      g.split_k = S; g.slab = G.slab; g.out_bf16 = nullptr; g.out_f32 = nullptr;
      KCHK(run_gemm(g, s));
      HIPCHK(kmb_ln_fwd_slabs_launch(G.slab, S, (size_t)R * d, h->pf(b_off), res, d, h->pf(g_off), h->pf(be_off), out, R, d,
                                     eps, s));
      return 0;
    }
    KmbGemm g = lin_fwd(in, K, h->wb(w_off), h->pf(b_off), R, d, K);
    g.residual = res; g.ld_res = d; g.out_bf16 = G.z; g.ld_out_bf16 = d;
    KCHK(run_gemm(g, s));
    HIPCHK(kmb_ln_fwd_launch(G.z, h->pf(g_off), h->pf(be_off), out, G.mean, G.rstd, R, d, eps, s));
    return 0;
  };
  // Fused form (csrc/decode.hip): six launches per layer, the LayerNorms folded into the consumers, weights read from
  // the fragment-order copies made by kmb_gen_begin.  Configurations gen_fused_eligible() rejects, and KMB_GEN_FUSED=0,
  // take the launch-per-operation path below.
  const char* fused_env = getenv("KMB_GEN_FUSED");   // read per call: tests compare the two paths in one process
  // the blocks work on 16-row tiles that each stream the layer's weights through L2: their time grows with the rows, while
  // the 128-row GEMM tiles of the launch-per-operation path amortise the weights (even at 1280 rows, faster below)
  const bool fused = !(fused_env && fused_env[0] == '0') && !G.wp.empty() && R <= 1024;
  if (fused) {
    const bf16_t* zin = G.x0;                       // layer input: normalised rows (layer 0) or pre-LayerNorm sums
    const float *lg = nullptr, *lb = nullptr;       // ... and the LayerNorm that turns them into the layer input
    auto block = [&](const KmbDecodeBlock& b) -> int {
      const char* why = kmb_decode_block_check(b);
      if (why) return fail("kmb_gen_step: %s", why);
      HIPCHK(kmb_decode_block_launch(b, s));
      return 0;
    };
#ifdef KMB_WITH_RESIDENT_DECODE
    // Resident form (tools/experiments/decode_resident.hip, round 5; experiment build `build.py --variant resident`, KMB_GEN_FUSED=2): ALL the layers in one launch
    // (KMB_GEN_LAYERS of them per launch), twelve co-resident workgroups per row tile behind counter barriers.  Bit-identical to the
    // six-launches-per-layer blocks and MEASURED SLOWER than them (12.0 against 10.3 ms per generate at batch 64 x 5 beams: an
    // in-kernel hand-off costs ~3.5 us where a kernel boundary costs ~1.7, and what a layer streams is bound by the CU's request
    // rate either way -- DESIGN.md section 4 "Generation", round 5; profiles/r05_generation_resident_kernel_stamps.md), so it is
    // opt-in; the blocks stay the default.
    const int Ld = h->cfg.decoder_layers;
    bool resident = fused_env && fused_env[0] == '2' && Ld > 0;
    KmbDecodeLayers A;
    if (resident) {
      memset(&A, 0, sizeof(A));
      A.n_layers = 1; A.x_in = G.x0; A.o = G.o; A.z = G.z; A.hh = G.hh; A.bars = G.bars; A.status = h->status;
      A.R = R; A.F = F; A.H = h->Hd; A.Tmax = G.Tmax; A.Tk = step + 1; A.S = G.S; A.ldc = Ld * 2 * d; A.kv_group = G.nb;
      A.key_mask = G.bt.attention_mask; A.mask_ld = G.S; A.eps = eps; A.q_scale = 0.125f;
      A.hist = G.use_hist ? G.hist[G.hcur] : nullptr;
      if (kmb_decode_layers_check(A) != nullptr) resident = false;
    }
    if (resident) {
      const char* lenv = getenv("KMB_GEN_LAYERS");
      int per = lenv && atoi(lenv) > 0 ? atoi(lenv) : Ld;
      if (per > KMB_DL_MAX_LAYERS) per = KMB_DL_MAX_LAYERS;
      for (int l0 = 0; l0 < Ld; l0 += per) {
        const int n = std::min(per, Ld - l0);
        A.n_layers = n;
        A.x_in = l0 == 0 ? G.x0 : G.z;
        for (int i = 0; i < n; ++i) {
          const int l = l0 + i;
          const LayerP& L = h->dec[l];
          KmbDecodeLayerP& P = A.L[i];
          P.Wqkv = G.wp[(size_t)l * 6 + 0]; P.Wo = G.wp[(size_t)l * 6 + 1]; P.Wcq = G.wp[(size_t)l * 6 + 2];
          P.Wco = G.wp[(size_t)l * 6 + 3]; P.W1 = G.wp[(size_t)l * 6 + 4]; P.W2 = G.wp[(size_t)l * 6 + 5];
          P.bqkv = h->pf(L.sa.qkv_b); P.bo = h->pf(L.sa.o_b); P.bcq = h->pf(L.ca.qkv_b); P.bco = h->pf(L.ca.o_b);
          P.b1 = h->pf(L.fc1_b); P.b2 = h->pf(L.fc2_b);
          P.lnin_g = l == 0 ? nullptr : h->pf(h->dec[l - 1].ln_g); P.lnin_b = l == 0 ? nullptr : h->pf(h->dec[l - 1].ln_b);
          P.ln1_g = h->pf(L.sa.ln_g); P.ln1_b = h->pf(L.sa.ln_b); P.ln2_g = h->pf(L.ca.ln_g); P.ln2_b = h->pf(L.ca.ln_b);
          P.Kc = G.kc[G.cur][l]; P.Vc = G.vc[G.cur][l]; P.cK = G.ckv[l]; P.cV = G.ckv[l] + d;
        }
        const hipError_t le = kmb_decode_layers_launch(A, s);
        if (le != hipSuccess) return fail("kmb_gen_step: resident decoder-layers launch failed: %s", hipGetErrorString(le));
      }
      zin = G.z; lg = h->pf(h->dec[Ld - 1].ln_g); lb = h->pf(h->dec[Ld - 1].ln_b);
    }
#else
    const bool resident = false;   // (the resident decoder-layers kernel is an experiment build since round 6: measured 17 % slower)
#endif
    for (int l = 0; !resident && l < h->cfg.decoder_layers; ++l) {
      const LayerP& L = h->dec[l];
      KmbDecodeBlock b;
      memset(&b, 0, sizeof(b));
      b.kind = 1; b.in = zin; b.ld_in = d; b.gamma = lg; b.beta = lb; b.eps = eps; b.ln_out = lg ? G.x1 : nullptr;
      b.W = G.wp[(size_t)l * 6 + 0]; b.bias = h->pf(L.sa.qkv_b); b.R = R; b.K = d; b.N = 3 * d; b.out = G.o; b.ld_out = d;
      b.H = h->Hd; b.q_scale = 0.125f; b.Kc = G.kc[G.cur][l]; b.Vc = G.vc[G.cur][l]; b.Tmax = G.Tmax; b.ldc = d; b.Tk = step + 1;
      b.hist = G.use_hist ? G.hist[G.hcur] : nullptr;
      KCHK(block(b));
      const bf16_t* xres = lg ? G.x1 : zin;
      memset(&b, 0, sizeof(b));
      b.kind = 0; b.in = G.o; b.ld_in = d; b.W = G.wp[(size_t)l * 6 + 1]; b.bias = h->pf(L.sa.o_b); b.R = R; b.K = d; b.N = d;
      b.residual = xres; b.ld_res = d; b.out = G.z; b.ld_out = d;
      KCHK(block(b));
      memset(&b, 0, sizeof(b));
      b.kind = 2; b.in = G.z; b.ld_in = d; b.gamma = h->pf(L.sa.ln_g); b.beta = h->pf(L.sa.ln_b); b.eps = eps; b.ln_out = G.y;
      b.W = G.wp[(size_t)l * 6 + 2]; b.bias = h->pf(L.ca.qkv_b); b.R = R; b.K = d; b.N = d; b.out = G.o; b.ld_out = d;
      b.H = h->Hd; b.q_scale = 0.125f; b.Kc = G.ckv[l]; b.Vc = G.ckv[l] + d; b.Tmax = G.S; b.ldc = h->cfg.decoder_layers * 2 * d; b.Tk = G.S;
      b.kv_row = G.kv_row; b.key_mask = G.bt.attention_mask; b.mask_ld = G.S; b.kv_group = G.nb;   // kv_row[i] = i / nb
      KCHK(block(b));
      memset(&b, 0, sizeof(b));
      b.kind = 0; b.in = G.o; b.ld_in = d; b.W = G.wp[(size_t)l * 6 + 3]; b.bias = h->pf(L.ca.o_b); b.R = R; b.K = d; b.N = d;
      b.residual = G.y; b.ld_res = d; b.out = G.z; b.ld_out = d;
      KCHK(block(b));
      memset(&b, 0, sizeof(b));
      b.kind = 0; b.in = G.z; b.ld_in = d; b.gamma = h->pf(L.ca.ln_g); b.beta = h->pf(L.ca.ln_b); b.eps = eps; b.ln_out = G.y;
      b.W = G.wp[(size_t)l * 6 + 4]; b.bias = h->pf(L.fc1_b); b.R = R; b.K = d; b.N = F; b.act = 1; b.out = G.hh; b.ld_out = F;
      KCHK(block(b));
      memset(&b, 0, sizeof(b));
      b.kind = 0; b.in = G.hh; b.ld_in = F; b.W = G.wp[(size_t)l * 6 + 5]; b.bias = h->pf(L.fc2_b); b.R = R; b.K = F; b.N = d;
      b.residual = G.y; b.ld_res = d; b.out = G.z; b.ld_out = d;
      KCHK(block(b));
      zin = G.z; lg = h->pf(L.ln_g); lb = h->pf(L.ln_b);
    }
    G.last_x = nullptr; G.last_z = G.z; G.last_g = lg; G.last_b = lb;
    if (lg && logits_out) {   // the last LayerNorm feeds only the vocabulary projection
      HIPCHK(kmb_ln_fwd_launch(G.z, lg, lb, G.x1, G.mean, G.rstd, R, d, eps, s));
      x = G.x1;
      G.last_x = x; G.last_z = nullptr;
    }
    if (!lg) { G.last_x = G.x0; G.last_z = nullptr; }   // a decoder without layers: the embedding output
  }
  for (int l = 0; !fused && l < h->cfg.decoder_layers; ++l) {
    const LayerP& L = h->dec[l];
    KmbGemm g = lin_fwd(x, d, h->wb(L.sa.qkv_w), h->pf(L.sa.qkv_b), R, 3 * d, d);
    g.col_scale = 0.125f; g.col_scale_n = d; g.out_bf16 = G.qkv; g.ld_out_bf16 = 3 * d;
    KCHK(run_gemm(g, s));
    KmbAttnDecode a; memset(&a, 0, sizeof(a));
    a.Q = G.qkv; a.ldq = 3 * d; a.Kc = G.kc[G.cur][l]; a.Vc = G.vc[G.cur][l]; a.Tmax = G.Tmax; a.ldc = d;
    a.R = R; a.H = h->Hd; a.Tk = step + 1; a.O = G.o; a.ldo = d;
    // this step's key / value: attended to from the projection output and appended to the cache by the same launch
    a.new_k = G.qkv + d; a.new_v = G.qkv + 2 * d; a.ld_new = 3 * d; a.Kw = G.kc[G.cur][l]; a.Vw = G.vc[G.cur][l];
    a.hist = G.use_hist ? G.hist[G.hcur] : nullptr;
    HIPCHK(kmb_attn_decode_launch(a, s));
    KCHK(proj_ln(G.o, d, L.sa.o_w, L.sa.o_b, x, L.sa.ln_g, L.sa.ln_b, G.y));
    // cross attention over the cached encoder K|V of the row's batch item
    g = lin_fwd(G.y, d, h->wb(L.ca.qkv_w), h->pf(L.ca.qkv_b), R, d, d);
    g.col_scale = 0.125f; g.col_scale_n = d; g.out_bf16 = G.cq; g.ld_out_bf16 = d;
    KCHK(run_gemm(g, s));
    memset(&a, 0, sizeof(a));
    a.Q = G.cq; a.ldq = d; a.Kc = G.ckv[l]; a.Vc = G.ckv[l] + d; a.Tmax = G.S; a.ldc = h->cfg.decoder_layers * 2 * d; a.kv_row = G.kv_row;
    a.key_mask = G.bt.attention_mask; a.mask_ld = G.S; a.mask_row = G.kv_row;
    a.R = R; a.H = h->Hd; a.Tk = G.S; a.O = G.o; a.ldo = d;
    HIPCHK(kmb_attn_decode_launch(a, s));
    KCHK(proj_ln(G.o, d, L.ca.o_w, L.ca.o_b, G.y, L.ca.ln_g, L.ca.ln_b, G.y));   // in place: a lane rewrites only the chunks it read
    // FFN
    g = lin_fwd(G.y, d, h->wb(L.fc1_w), h->pf(L.fc1_b), R, F, d);
    g.act = 1; g.out_bf16 = G.hh; g.ld_out_bf16 = F;
    KCHK(run_gemm(g, s));
    KCHK(proj_ln(G.hh, F, L.fc2_w, L.fc2_b, G.y, L.ln_g, L.ln_b, xn));
    bf16_t* t = x; x = xn; xn = t;
  }
  if (!fused) { G.last_x = x; G.last_z = nullptr; }
  G.head_stats_blocks = 0; G.head_stats_for = nullptr;
  if (logits_out) {
    KmbGemm g = lin_fwd(x, d, h->wb(h->shared), h->flb, R, h->V, d);
    g.out_f32 = logits_out; g.ld_out_f32 = h->Vpad;
    // KMB_GEN_HEAD_STATS=0: the projection without its statistics epilogue, the beam step in two launches over the logits (read per call:
    // tests compare the two in one process)
    const char* hs_env = getenv("KMB_GEN_HEAD_STATS");
    const bool want_stats = !(hs_env && hs_env[0] == '0');
    KCHK(run_vocab_gemm(g, s, want_stats ? G.head_stats : nullptr, want_stats ? &G.head_stats_blocks : nullptr));
    if (G.head_stats_blocks > 0) G.head_stats_for = logits_out;
  }
  return 0;
}

int kmb_gen_reorder(kmb_handle* h, const int32_t* beam_idx, int step, void* stream);

// The beam step of the decode loop on the logits of the last kmb_gen_step (mixins.py:386-417 via transformers 3.0.2
// _generate_beam_search: log_softmax + beam score, the 2 * num_beams best per batch item, the next step's beams): kmb_beam_step's
// arguments and outputs.  When that step's vocabulary projection left its per-block statistics (all-rows kernel, 257 .. 320 beam rows),
// ONE launch selects from them; otherwise kmb_beam_step's two launches over the logits.
// reorder_step >= 0: also kmb_gen_reorder(next_beam_idx, reorder_step) (_reorder_cache, mixins.py:419-434) -- with the history index
// (the default) by the launch that has just chosen the beams, no launch of its own.
int kmb_gen_beam_step(kmb_handle* h, const float* logits, int ld, int num_beams, const float* add, int force_token, int ban_token, int k,
                      int32_t* out, int eos_token, float* next_scores, int64_t* next_tokens, int32_t* next_beam_idx, float* scratch,
                      int64_t scratch_floats, int reorder_step, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  auto& G = h->gen;
  if (!G.active) return fail("kmb_gen_beam_step: call kmb_gen_begin first");
  if (!logits || !out || !next_scores || !next_tokens || !next_beam_idx) return fail("kmb_gen_beam_step: missing tensor");
  if (num_beams != G.nb) return fail("kmb_gen_beam_step: num_beams %d, kmb_gen_begin had %d", num_beams, G.nb);
  if (reorder_step >= G.Tmax) return fail("kmb_gen_beam_step: reorder_step %d outside the cache (Tmax=%d)", reorder_step, G.Tmax);
  const bool fold = reorder_step >= 0 && G.use_hist;
  KmbHistGather hg{G.hist[G.hcur], G.hist[G.hcur ^ 1], G.Tmax, reorder_step + 1};
  // a reorder means another decode step follows, at position reorder_step + 1, on the tokens chosen here: the same launch embeds them
  // (G.x0 is free: the step's layers have run).  KMB_GEN_FOLD_EMBED=0: kmb_gen_step's own embedding launch, as before round 6.
  const char* fe_env = getenv("KMB_GEN_FOLD_EMBED");
  const int d = h->d;
  const bool embed = reorder_step >= 0 && reorder_step + 1 < G.Tmax && !(fe_env && fe_env[0] == '0') && (d & 7) == 0 && d > 512 && d <= 1024;
  KmbEmbedNext en;
  if (embed) {
    en.E = h->pf(h->shared); en.prow = h->pf(h->dec_pos) + (size_t)(h->cfg.extra_pos_embeddings + reorder_step + 1) * d;
    en.gamma = h->pf(h->dec_lne_g); en.beta = h->pf(h->dec_lne_b); en.y = G.x0;
    en.scale = h->cfg.scale_embedding ? sqrtf((float)d) : 1.f; en.D = d; en.eps = h->cfg.layer_norm_eps; en.V = h->V;
  }
  G.x0_step = -1; G.x0_tokens = nullptr;
  hipError_t e = hipErrorNotSupported;
  if (force_token < 0 && G.head_stats_blocks > 0 && G.head_stats_for == logits)
    e = kmb_beam_step_stats_launch(logits, ld, h->V, G.B, num_beams, add, force_token, ban_token, k, out, eos_token, next_scores,
                                   next_tokens, next_beam_idx, G.head_stats, G.head_stats_blocks, s, fold ? &hg : nullptr,
                                   embed ? &en : nullptr);
  if (e == hipErrorNotSupported)
    e = kmb_beam_step_launch(logits, ld, h->V, G.B, num_beams, add, force_token, ban_token, k, out, eos_token, next_scores, next_tokens,
                             next_beam_idx, scratch, scratch_floats > 0 ? (size_t)scratch_floats : 0, s, fold ? &hg : nullptr,
                             embed ? &en : nullptr);
  if (e == hipErrorNotSupported) return fail("kmb_gen_beam_step: unsupported shape (k <= 16, num_beams <= 16, num_beams * k <= 256)");
  HIPCHK(e);
  if (embed) { G.x0_step = reorder_step + 1; G.x0_tokens = next_tokens; }
  if (fold) {
    G.hcur ^= 1;
    if (G.nb == 1) {   // independent rows: the row -> cross-attention item table follows (kmb_gen_reorder)
      int32_t* other = G.kv_row == G.kv_row_base ? G.kv_row_base + G.R : G.kv_row_base;
      HIPCHK(kmb_gather_i32_launch(G.kv_row, next_beam_idx, other, G.R, s));
      G.kv_row = other;
    }
  } else if (reorder_step >= 0) {
    return kmb_gen_reorder(h, next_beam_idx, reorder_step, stream);   // KMB_GEN_HIST=0: the physical reorder
  }
  return 0;
}

int kmb_gen_last_hidden(kmb_handle* h, kmb_bf16* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  auto& G = h->gen;
  if (!G.active) return fail("kmb_gen_last_hidden: call kmb_gen_begin first");
  if (!out) return fail("kmb_gen_last_hidden: out is required");
  if (G.last_x) {
    HIPCHK(hipMemcpyAsync(out, G.last_x, (size_t)G.R * h->d * sizeof(bf16_t), hipMemcpyDeviceToDevice, s));
  } else if (G.last_z && G.last_g) {
    HIPCHK(kmb_ln_fwd_launch(G.last_z, G.last_g, G.last_b, out, G.mean, G.rstd, G.R, h->d, h->cfg.layer_norm_eps, s));
  } else {
    return fail("kmb_gen_last_hidden: no kmb_gen_step has run since kmb_gen_begin");
  }
  return 0;
}

int kmb_gen_reorder(kmb_handle* h, const int32_t* beam_idx, int step, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  auto& G = h->gen;
  if (!G.active) return fail("kmb_gen_reorder: call kmb_gen_begin first");
  const int d = h->d;
  if (G.use_hist) {   // permute the history index, not the caches
    HIPCHK(kmb_gather_hist_launch(G.hist[G.hcur], beam_idx, G.hist[G.hcur ^ 1], G.R, G.Tmax, step + 1, s));
    G.hcur ^= 1;
    if (G.nb == 1) {   // independent rows: the row -> cross-attention item table follows (see below)
      int32_t* other = G.kv_row == G.kv_row_base ? G.kv_row_base + G.R : G.kv_row_base;
      HIPCHK(kmb_gather_i32_launch(G.kv_row, beam_idx, other, G.R, s));
      G.kv_row = other;
    }
    return 0;
  }
  const int row_bytes = (step + 1) * d * (int)sizeof(bf16_t);
  const size_t stride = (size_t)G.Tmax * d * sizeof(bf16_t);
  // every layer's K and V cache in one launch per 16 buffers (12 launches -> 1 for a 6-layer decoder)
  const void* src[16];
  void* dst[16];
  int n = 0;
  for (int l = 0; l < h->cfg.decoder_layers; ++l) {
    for (int kv = 0; kv < 2; ++kv) {
      src[n] = kv ? (const void*)G.vc[G.cur][l] : (const void*)G.kc[G.cur][l];
      dst[n] = kv ? (void*)G.vc[G.cur ^ 1][l] : (void*)G.kc[G.cur ^ 1][l];
      if (++n == 16) {
        HIPCHK(kmb_gather_rows_multi_launch(src, dst, n, beam_idx, G.R, row_bytes, stride, s));
        n = 0;
      }
    }
  }
  if (n) HIPCHK(kmb_gather_rows_multi_launch(src, dst, n, beam_idx, G.R, row_bytes, stride, s));
  G.cur ^= 1;
  if (G.nb == 1) {
    // rows are independent sequences (the cached forward of src/model/model.py:384-397 with caller-expanded rows):
    // _reorder_cache (mixins.py:419-434) also permutes the encoder side, here the row -> cross-attention item table.
    // (With num_beams > 1 a beam search only permutes rows inside a batch item and the table is unchanged.)
    int32_t* other = G.kv_row == G.kv_row_base ? G.kv_row_base + G.R : G.kv_row_base;
    HIPCHK(kmb_gather_i32_launch(G.kv_row, beam_idx, other, G.R, s));
    G.kv_row = other;
  }
  return 0;
}

}  // extern "C"
