// Host-side engine: chains the gfx950 kernels into model/bert.py BertModel forward / backward.
// No device code here; everything is asynchronous launches on the caller's stream (graph-capturable).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/mmt_hip.h"

namespace {

enum { SITE_EMB = 0, SITE_PROBS = 1, SITE_ATTN_OUT = 2, SITE_FFN_OUT = 3 };
inline uint32_t site_key(int layer, int site) { return 0x5eed0000u + (uint32_t)layer * 16u + (uint32_t)site; }
inline uint32_t thr16_of(float p) { int t = (int)(p * 65536.0f + 0.5f); return t < 0 ? 0u : (uint32_t)t; }
inline float scale_of(uint32_t thr) { return thr ? 1.0f / (1.0f - (float)thr / 65536.0f) : 1.0f; }

struct LayerWs {
  char *qkv, *ctx, *a16, *hpre, *g, *h16;
  float *lse, *z1, *mean1, *rstd1, *a32, *z2, *mean2, *rstd2, *h32;
};
// In the compact last layer only the QKV weight gradient still contracts over every token row: its tiles are split
// TAIL_WSPLIT ways over the rows so that they do not outlast the (short) other items of the grouped launch.
constexpr int TAIL_WSPLIT = 4;
// Compact buffers of the last layer when only `out_rows` are needed (rows_c = B * n_out_per_sample).
struct TailWs {
  char *ctx, *a16, *hpre, *g, *dy2, *dy, *dhpre, *dctx;
  float *lse, *res32, *z1, *mean1, *rstd1, *a32, *z2, *mean2, *rstd2, *dcur, *dz, *dA, *delta;
  int32_t* rowidx;
  float* slabs;  // split-K partials of the skinny long-K GEMMs
  float *wslab, *bslab;  // split partials of the last layer's QKV weight / bias gradient
  int cap;  // rows the buffers can hold
};
struct Ws {
  TailWs t;
  float *z0, *mean0, *rstd0, *h32_in;
  char* h16_in;
  LayerWs layer[64];
  float *dz, *dA, *delta, *ln_partials[2 * 64 + 1], *table_scratch[2];
  // dy / dy2 / dhpre / dqkv are what the weight gradients of a layer read: two sets, used by even / odd layers, so that
  // the weight gradients of layer l may still be running (MMT_FORK_WGRAD) while layer l-1 produces its own
  char *dy_[2], *dy2_[2], *dhpre_[2], *dqkv_[2], *dctx;
  size_t attn_work_words;
  int32_t* attn_work;  // block order of the attention backward for the batch in flight (attn_sched.h; written by the forward)
  float* kslab;  // split-K partial slabs of the main path's K = intermediate, N = hidden GEMMs (up to 4 x [R, d])
  size_t bytes;
};

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

void layout(const MmtBertModel* m, int R, char* base, Ws* w) {
  const size_t d = m->hidden, I = m->inter, H = m->heads;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
  w->z0 = (float*)take(R * d * 4); w->mean0 = (float*)take(R * 4); w->rstd0 = (float*)take(R * 4);
  w->h32_in = (float*)take(R * d * 4); w->h16_in = take(R * d * 2);
  for (int l = 0; l < m->layers; ++l) {
    LayerWs& L = w->layer[l];
    L.qkv = take(R * 3 * d * 2); L.ctx = take(R * d * 2); L.lse = (float*)take(R * H * 4);
    L.z1 = (float*)take(R * d * 4); L.mean1 = (float*)take(R * 4); L.rstd1 = (float*)take(R * 4);
    L.a32 = (float*)take(R * d * 4); L.a16 = take(R * d * 2);
    L.hpre = take(R * I * 2); L.g = take(R * I * 2);
    L.z2 = (float*)take(R * d * 4); L.mean2 = (float*)take(R * 4); L.rstd2 = (float*)take(R * 4);
    L.h32 = (float*)take(R * d * 4); L.h16 = take(R * d * 2);
  }
  w->dz = (float*)take(R * d * 4); w->dA = (float*)take(R * d * 4);
  w->dy_[0] = take(R * d * 2); w->dy2_[0] = take(R * d * 2); w->dhpre_[0] = take(R * I * 2); w->dctx = take(R * d * 2);
  w->dqkv_[0] = take(R * 3 * d * 2);
  w->delta = (float*)take(R * (d / 64) * 4);  // dO * O sums per 64-column group (attention backward)
  size_t ln_nb = ((size_t)R + 15) / 16;  // blocks of mmt_ln_bwd: rows/16, or rows/4 when rows <= 2048
  const size_t small_nb = ((size_t)R + 3) / 4 < 512 ? ((size_t)R + 3) / 4 : 512;
  if (ln_nb < small_nb) ln_nb = small_nb;
  for (int i = 0; i < 2 * m->layers + 1; ++i) w->ln_partials[i] = (float*)take(ln_nb * 3 * d * 4);
  const int pos_v = m->max_pos > 128 ? 0 : m->max_pos;  // larger position tables take mmt_table_grad_direct (no scratch)
  const int vmax = m->type_vocab > pos_v ? m->type_vocab : pos_v;
  {  // tail buffers: B*M read-out rows are at most a quarter of the token rows for T >= 3 (else: full path)
    const size_t C = (size_t)mmt_bert_tail_capacity(R);
    TailWs& t = w->t;
    t.cap = (int)C;
    t.ctx = take(C * d * 2); t.a16 = take(C * d * 2); t.hpre = take(C * I * 2); t.g = take(C * I * 2);
    t.dy2 = take(C * d * 2); t.dy = take(C * d * 2); t.dhpre = take(C * I * 2); t.dctx = take(C * d * 2);
    t.lse = (float*)take(C * H * 4); t.res32 = (float*)take(C * d * 4); t.z1 = (float*)take(C * d * 4);
    t.mean1 = (float*)take(C * 4); t.rstd1 = (float*)take(C * 4); t.a32 = (float*)take(C * d * 4);
    t.z2 = (float*)take(C * d * 4); t.mean2 = (float*)take(C * 4); t.rstd2 = (float*)take(C * 4);
    t.dcur = (float*)take(C * d * 4); t.dz = (float*)take(C * d * 4); t.dA = (float*)take(C * d * 4);
    t.delta = (float*)take(C * (d / 64) * 4); t.rowidx = (int32_t*)take(C * 4);
    t.wslab = (float*)take((size_t)TAIL_WSPLIT * 3 * d * d * 4); t.bslab = (float*)take((size_t)TAIL_WSPLIT * 3 * d * 4);
    t.slabs = (float*)take((size_t)mmt_gemm_nt_splitk_workspace_floats((int)C, (int)d, (int)I) * 4);
  }
  for (int i = 0; i < 2; ++i) w->table_scratch[i] = (float*)take((size_t)mmt_table_grad_scratch_floats(vmax, (int)d) * 4);
  // second set of the weight-gradient operands (odd layers under MMT_FORK_WGRAD), behind everything else: the buffers of
  // the serial path keep the addresses (and the cache-set relationships) they had before forking existed
  w->dy_[1] = take(R * d * 2); w->dy2_[1] = take(R * d * 2); w->dhpre_[1] = take(R * I * 2); w->dqkv_[1] = take(R * 3 * d * 2);
  w->kslab = (float*)take((size_t)4 * R * d * 4);
  // B * ceil(S / 64) <= R / 64 + B tile slots per role and head, and a schedule is only built for B * H <= 2048
  w->attn_work_words = (size_t)2 * ((size_t)R / 64 + (R < 2048 ? (size_t)R : 2048)) * H * 4;
  w->attn_work = (int32_t*)take(w->attn_work_words * 4);
  w->bytes = off;
}

int check_model(const MmtBertModel* m, const MmtBertBatch* b) {
  if (!m || !b || !m->layer) return MMT_ERR_ARG;
  if (m->layers <= 0 || m->layers > 64) return MMT_ERR_ARG;
  if (m->hidden != m->heads * 128 && m->hidden != m->heads * 64) return MMT_ERR_ARG;  // head dim 128 (video BERT) or 64 (BERT-base)
  if (m->hidden % 256 || m->hidden > 1024 || m->inter % 128) return MMT_ERR_ARG;
  if (b->rows <= 0 || b->rows > b->rows_alloc || b->rows_alloc % MMT_ROW_ALIGN) return MMT_ERR_ARG;
  if (!b->features || !b->type_ids || !b->mask_bias) return MMT_ERR_ARG;
  return 0;
}

// Profiling probes (bench.py): HIP events recorded on the launch stream around ONE launch per step of the three kernel
// families that lead the rocprof time table -- site 0: FFN up-projection GEMM (+GELU) of layer 0, site 1: FFN
// down-projection GEMM (N = hidden, K = intermediate; + bias/dropout/residual) of layer 0, site 2: the grouped
// weight-gradient launch of layer 0, sites 3 / 4: the attention forward / backward launch of layer 0 -- until the armed
// event pairs of a site are used up.
enum { PROBE_SITES = 5 };
hipEvent_t* g_probe_start[PROBE_SITES] = {};
hipEvent_t* g_probe_stop[PROBE_SITES] = {};
int g_probe_n[PROBE_SITES] = {}, g_probe_i[PROBE_SITES] = {};

struct ProbeScope {
  int site; hipStream_t s; bool on;
  ProbeScope(int site_, bool cond, void* stream) : site(site_), s((hipStream_t)stream) {
    on = cond && g_probe_i[site] < g_probe_n[site];
    if (on) (void)hipEventRecord(g_probe_start[site][g_probe_i[site]], s);
  }
  ~ProbeScope() {
    if (on) (void)hipEventRecord(g_probe_stop[site][g_probe_i[site]++], s);
  }
};

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// rows of the compact last layer, or 0 when the full path must be used
int tail_rows(const MmtBertBatch* b, const Ws& w) {
  if (!b->out_rows || b->n_out_per_sample <= 0) return 0;
  const long n = (long)b->batch * b->n_out_per_sample;
  return (n > 0 && n <= w.t.cap) ? (int)n : 0;
}

// ---- forked launches (MmtBertBatch.fork) ----
// "weight gradients of layer l have been issued" events, one per layer parity and workspace: layer l-2 overwrites the
// buffers layer l's weight gradients read, so `stream` waits for that event first.
struct DoneEvents { const void* ws; hipEvent_t ev[2]; unsigned long long cap[2]; };  // cap: capture the record happened in (0: none)
DoneEvents g_done[16];
int g_done_n = 0;
std::mutex g_done_mu;
unsigned long long capture_id(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(s, &st, &id) != hipSuccess || st != hipStreamCaptureStatusActive) return 0;
  return id;
}
DoneEvents* done_entry(const void* ws);
hipEvent_t* done_events(const void* ws) {
  DoneEvents* e = done_entry(ws);
  return e ? e->ev : nullptr;
}
DoneEvents* done_entry(const void* ws) {
  std::lock_guard<std::mutex> lock(g_done_mu);
  for (int i = 0; i < g_done_n; ++i)
    if (g_done[i].ws == ws) return &g_done[i];
  DoneEvents& e = g_done[g_done_n < 16 ? g_done_n++ : 15];  // (more than 16 live workspaces: the last slot is recycled)
  e.ws = ws;
  for (int p = 0; p < 2; ++p) {
    if (!e.ev[p] && hipEventCreateWithFlags(&e.ev[p], hipEventDisableTiming) != hipSuccess) return nullptr;
    e.cap[p] = 0;
  }
  return &e;
}

}  // namespace

extern "C" int mmt_probe_arm_site(int site, void** start_events, void** stop_events, int n) {
  if (site < 0 || site >= PROBE_SITES) return MMT_ERR_ARG;
  g_probe_start[site] = (hipEvent_t*)start_events;
  g_probe_stop[site] = (hipEvent_t*)stop_events;
  g_probe_n[site] = (start_events && stop_events) ? n : 0;
  g_probe_i[site] = 0;
  return 0;
}
extern "C" int mmt_probe_arm(void** start_events, void** stop_events, int n) {
  return mmt_probe_arm_site(0, start_events, stop_events, n);
}
extern "C" int mmt_probe_count_site(int site) { return site >= 0 && site < PROBE_SITES ? g_probe_i[site] : 0; }
extern "C" int mmt_probe_count(void) { return g_probe_i[0]; }

extern "C" int mmt_bert_tail_capacity(int rows_alloc) { return (int)((((size_t)rows_alloc / 4) + 255) & ~(size_t)255); }

extern "C" int64_t mmt_bert_workspace_bytes(const MmtBertModel* m, int rows_alloc) {
  if (!m || rows_alloc <= 0 || m->layers > 64) return MMT_ERR_ARG;
  Ws w;
  layout(m, rows_alloc, nullptr, &w);
  return (int64_t)w.bytes;
}

// N = hidden GEMMs with a long K on SHORT batches (the text tower: ~1000 token rows x 768): too few output tiles for
// 256 CUs and a 36..48-step dependent K loop per tile -> split K over 2..4 blocks per tile (partials in the tail's slab
// workspace, which is idle outside the tail layer) and reduce in the epilogue kernel.  Measured 30 -> 17 us (K = 3072),
// 24 -> 15.5 us (K = 2304) at 960 rows; not worth it at K = 768.  The video side (thousands of rows) never takes it.
static int gemm_hidden(const Ws& w, int rows, int d, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                       int64_t ldc, int K, int epi, const MmtEpilogue* e, const int32_t* nr, void* stream) {
  const int tiles = ((rows + 127) / 128) * (d / 64), ksteps = K / 64;
  if (tiles < 160 && ksteps >= 32 && rows <= 4 * w.t.cap)
    return mmt_gemm_nt_splitk_ex(A, lda, B, ldb, C, ldc, rows, d, K, epi, e, w.t.slabs, ksteps >= 48 ? 4 : 2, 0, nr, 0, stream);
  return mmt_gemm_nt_bf16(A, lda, B, ldb, C, ldc, rows, d, K, epi, e, nr, stream);
}

// K = intermediate, N = hidden on thousands of rows: one 128x64 tile per CU walks 48 dependent K-steps at the CU's
// L2 -> LDS ingest limit (24 KB per step).  Lab switch MMT_SPLITK_FFN = 10 * splits + wide (0 = off, the default): the GEMM
// split over K with the partial slabs summed by the LayerNorm pass that follows anyway (fwd: + bias, dropout, residual;
// bwd: + residual gradient).  Same-box A/B of the whole step (3 alternations, r04): off 1.2864 ms; 2 splits of 128x64
// tiles 1.2912; 2 splits of 128x128 tiles 1.3245; 4 splits of 256x128 tiles 1.2808 -- the slab traffic (4 x 7.3 MB written
// and read back per GEMM) eats what the shorter K-chains win.  Parity-tested (tests/test_cenet_gpu.py under the switch).
static int splitk_ffn_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MMT_SPLITK_FFN");
    mode = e ? atoi(e) : 0;
  }
  return mode;
}
static bool splitk_ffn(const Ws&, int rows, int d, int K) {
  return splitk_ffn_mode() > 0 && rows >= 2048 && K >= 2048 && d <= 512;
}

// r06: the short-batch case of gemm_hidden (the text tower: a few hundred live rows, K = intermediate) already runs split-K;
// its slab-summing epilogue launch and the LayerNorm launch behind it are ONE launch when the LayerNorm reads the slabs
// itself (mmt_splitk_ln_fwd_ex / mmt_ln_bwd_slabs_ex: + bias, dropout, residual resp. + residual gradient) -- two graph
// nodes less per layer, forward and backward.  -> splits (0: not this case).  MMT_SPLITK_LN=0 restores the r05 launches.
static int small_splitk_ln(const Ws& w, int rows, int d, int K) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MMT_SPLITK_LN");
    on = e ? atoi(e) : 1;
  }
  const int tiles = ((rows + 127) / 128) * (d / 64), ksteps = K / 64;
  if (!on || !(tiles < 160 && ksteps >= 32 && rows <= 4 * w.t.cap)) return 0;
  return ksteps >= 48 ? 4 : 2;
}

// r05: BertSelfOutput's projection + LayerNorm as one launch where the hidden size is the 512 of the video BERT (a block of
// gemm_ln.hip owns 32 whole rows of 512 columns).  OPT-IN (MMT_FUSE_OUT_LN=1): it removes three graph nodes and three reads
// of z per step, but every block streams the whole 512 KiB weight through its CU's ~21 B/clk ingest -- same-box A/B of the
// whole step, three alternations: 1.2718 / 1.2719 / 1.2738 ms with the GEMM + LayerNorm pair, 1.2851 / 1.2863 / 1.2871 with
// the fused launch.
static bool fuse_out_ln(int d) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MMT_FUSE_OUT_LN");
    on = e ? atoi(e) : 0;
  }
  return on && d == 512;
}

// the attention backward's block order (attn_sched.h) exists for packed batches whose (sample, head) pairs split over 8 XCDs
static const int32_t* attn_work_of(const MmtBertModel* m, const MmtBertBatch* b, const Ws& w) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MMT_ATTN_SCHED");
    on = e ? atoi(e) : 1;
  }
  const long bh = (long)b->batch * m->heads;
  if (!on || !b->cu_seqlens || bh % 8 || bh > 2048 || m->heads > 255) return nullptr;
  if ((size_t)mmt_attn_schedule_words(b->batch, b->seq, m->heads) > w.attn_work_words) return nullptr;
  return w.attn_work;
}

extern "C" int mmt_bert_forward(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* out_last,
                                int training, void* stream) {
  TRY(check_model(m, b));
  if (!ws || !out_last) return MMT_ERR_ARG;
  Ws w;
  layout(m, b->rows_alloc, (char*)ws, &w);
  const int d = m->hidden, I = m->inter, rows = b->rows;
  const uint32_t th = training ? thr16_of(m->p_hidden) : 0u, ta = training ? thr16_of(m->p_attn) : 0u;
  const float sh = scale_of(th), sa = scale_of(ta);
  const int lh = b->n_rows_dev ? b->live_rows_hint : 0;  // the host's live row count (tile choice only: gemm.hip select_tile)
  const float qk_scale = m->hidden == m->heads * 128 ? 0.08838834764831845f : 0.125f;  // 1/sqrt(head dim)

  TRY(mmt_embed_ln_fwd_sched(b->features, b->type_ids, b->pos_ids, m->type_emb, m->pos_emb, w.z0, m->emb_ln_g,
                       m->emb_ln_b, m->ln_eps, w.h32_in, w.h16_in, w.mean0, w.rstd0, rows, d, b->n_rows_dev,
                       b->row_index, site_key(0, SITE_EMB), th, sh, b->seed_dev, b->cu_seqlens, b->batch, m->heads, (b->seq + 63) / 64,
                             const_cast<int32_t*>(attn_work_of(m, b, w)), stream));
  const float* hin32 = w.h32_in;
  const char* hin16 = w.h16_in;
  const int nc = tail_rows(b, w);
  for (int l = 0; l < m->layers; ++l) {
    const MmtBertLayer& P = m->layer[l];
    LayerWs& L = w.layer[l];
    MmtEpilogue e = {}; e.live_rows_hint = lh;
    e.bias = P.bqkv;
    TRY(mmt_gemm_nt_bf16(hin16, d, P.wqkv, d, L.qkv, 3 * d, rows, 3 * d, d, MMT_EPI_BIAS_BF16, &e, b->n_rows_dev, stream));
    if (nc && l == m->layers - 1) {
      // ---- last layer, read-out rows only: everything after the K/V projection is row-wise, so only the nc rows
      // the caller reads are computed (as queries against ALL keys).  RNG coordinates travel in t.rowidx, so the
      // dropout masks are the ones the full path would draw. ----
      TailWs& t = w.t;
      TRY(mmt_attn_fwd_rows(L.qkv, b->cu_seqlens, b->mask_bias, b->out_rows, b->n_out_per_sample, t.ctx, t.lse, b->batch,
                            b->seq, m->heads, d, qk_scale, site_key(l, SITE_PROBS), ta, sa, b->seed_dev, b->row_index, stream));
      // The split-K partial slabs of the two N = hidden GEMMs are summed by the LayerNorm pass that follows them, which
      // also applies bias, dropout and the residual (gathered straight from the previous layer's output rows for the
      // attention block): one launch instead of reduce-epilogue + LayerNorm (+ a row gather).
      int sp = 0;
      int64_t sstride = 0;
      TRY(mmt_gemm_splitk_geometry(nc, d, d, 0, &sp, &sstride));
      TRY(mmt_gemm_nt_splitk_ex(t.ctx, d, P.wo, d, nullptr, d, nc, d, d, MMT_EPI_F32, nullptr, t.slabs, 0, 0, nullptr, 1, stream));
      TRY(mmt_splitk_ln_fwd(t.slabs, sp, sstride, P.bo, hin32, b->out_rows, nullptr, b->row_index, t.rowidx,
                            site_key(l, SITE_ATTN_OUT), th, sh, b->seed_dev, t.z1, P.ln1_g, P.ln1_b, m->ln_eps, t.a32, t.a16,
                            t.mean1, t.rstd1, nc, d, stream));
      e = {}; e.live_rows_hint = lh;
      e.bias = P.b1; e.out2 = t.g; e.ldout2 = I;
      TRY(mmt_gemm_nt_bf16(t.a16, d, P.w1, d, t.hpre, I, nc, I, d, MMT_EPI_BIAS_GELU, &e, nullptr, stream));
      TRY(mmt_gemm_splitk_geometry(nc, d, I, 0, &sp, &sstride));
      TRY(mmt_gemm_nt_splitk_ex(t.g, I, P.w2, I, nullptr, d, nc, d, I, MMT_EPI_F32, nullptr, t.slabs, 0, 0, nullptr, 1, stream));
      // the nc read-out rows are returned COMPACT: out_last[i] = sequence_output[out_rows[i]], i < nc
      TRY(mmt_splitk_ln_fwd(t.slabs, sp, sstride, P.b2, t.a32, nullptr, t.rowidx, nullptr, nullptr, site_key(l, SITE_FFN_OUT), th,
                            sh, b->seed_dev, t.z2, P.ln2_g, P.ln2_b, m->ln_eps, out_last, nullptr, t.mean2, t.rstd2, nc, d,
                            stream));
      break;
    }
    {
      ProbeScope probe(3, l == 0, stream);
      TRY(mmt_attn_fwd(L.qkv, b->cu_seqlens, b->mask_bias, L.ctx, L.lse, b->batch, b->seq, m->heads, d, qk_scale,
                       site_key(l, SITE_PROBS), ta, sa, b->seed_dev, b->row_index, stream));
    }
    if (fuse_out_ln(d)) {
      // attention output projection + dropout + residual + LayerNorm in ONE launch (gemm_ln.hip: a block owns 32 whole rows)
      TRY(mmt_gemm_nt_ln_fwd(L.ctx, d, P.wo, d, P.bo, hin32, d, b->row_index, site_key(l, SITE_ATTN_OUT), th, sh, b->seed_dev,
                             L.z1, P.ln1_g, P.ln1_b, m->ln_eps, L.a32, L.a16, L.mean1, L.rstd1, rows, d, d, b->n_rows_dev, stream));
    } else {
    e = {}; e.live_rows_hint = lh;
    e.bias = P.bo; e.res = hin32; e.ldres = d; e.row_index = b->row_index; e.seed_dev = b->seed_dev;
    e.drop_key = site_key(l, SITE_ATTN_OUT); e.drop_thr16 = th; e.drop_scale = sh;
    TRY(mmt_gemm_nt_bf16(L.ctx, d, P.wo, d, L.z1, d, rows, d, d, MMT_EPI_BIAS_DROP_RES, &e, b->n_rows_dev, stream));
    TRY(mmt_ln_fwd(L.z1, P.ln1_g, P.ln1_b, m->ln_eps, L.a32, L.a16, L.mean1, L.rstd1, rows, d, b->n_rows_dev, stream));
    }
    e = {}; e.live_rows_hint = lh;
    e.bias = P.b1; e.out2 = L.g; e.ldout2 = I;
    {
      ProbeScope probe(0, l == 0, stream);
      TRY(mmt_gemm_nt_bf16(L.a16, d, P.w1, d, L.hpre, I, rows, I, d, MMT_EPI_BIAS_GELU, &e, b->n_rows_dev, stream));
    }
    e = {}; e.live_rows_hint = lh;
    e.bias = P.b2; e.res = L.a32; e.ldres = d; e.row_index = b->row_index; e.seed_dev = b->seed_dev;
    e.drop_key = site_key(l, SITE_FFN_OUT); e.drop_thr16 = th; e.drop_scale = sh;
    float* hout32 = (l == m->layers - 1) ? out_last : L.h32;
    const int small_sp = small_splitk_ln(w, rows, d, I);
    if (splitk_ffn(w, rows, d, I) || small_sp) {
      const int mode = small_sp ? 10 * small_sp : splitk_ffn_mode();
      float* slabs = small_sp ? w.t.slabs : w.kslab;
      int sp = 0;
      int64_t sstride = 0;
      TRY(mmt_gemm_splitk_geometry(rows, d, I, mode / 10, &sp, &sstride));
      {
        ProbeScope probe(1, l == 0, stream);
        TRY(mmt_gemm_nt_splitk_ex(L.g, I, P.w2, I, nullptr, d, rows, d, I, MMT_EPI_F32, nullptr, slabs, mode / 10, mode % 10,
                                  b->n_rows_dev, 1, stream));
      }
      TRY(mmt_splitk_ln_fwd_ex(slabs, sp, sstride, P.b2, L.a32, nullptr, nullptr, b->row_index, nullptr,
                               site_key(l, SITE_FFN_OUT), th, sh, b->seed_dev, L.z2, P.ln2_g, P.ln2_b, m->ln_eps, hout32, L.h16,
                               L.mean2, L.rstd2, rows, d, b->n_rows_dev, stream));
      hin32 = hout32;
      hin16 = L.h16;
      continue;
    }
    {
      ProbeScope probe(1, l == 0, stream);
      TRY(gemm_hidden(w, rows, d, L.g, I, P.w2, I, L.z2, d, I, MMT_EPI_BIAS_DROP_RES, &e, b->n_rows_dev, stream));
    }
    TRY(mmt_ln_fwd(L.z2, P.ln2_g, P.ln2_b, m->ln_eps, hout32, L.h16, L.mean2, L.rstd2, rows, d, b->n_rows_dev, stream));
    hin32 = hout32;
    hin16 = L.h16;
  }
  return 0;
}

// Layers l_hi .. l_lo (descending); the embedding stage runs with layer 0.  A full backward is (layers-1, 0); a caller
// that wants to start reducing a layer's gradients over ranks while the next layers still run calls it range by range
// (the buffers that carry the running gradient between calls are determined by the layer index alone).
//
// b->fork / b->side_stream: the launches nothing in the remaining backward depends on -- weight gradients, LayerNorm /
// table reductions -- go to the caller's second stream, ordered by events (graph edges under capture), and run under the
// input-gradient chain of the layers below.
extern "C" int mmt_bert_backward_range(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* dlast,
                                       float* dfeatures, int training, int l_hi, int l_lo, void* stream) {
  TRY(check_model(m, b));
  const bool embed_only = l_hi == -1 && l_lo == -1;  // just the embedding stage (after a MMT_RANGE_LAYERS_ONLY call)
  if (!ws || !dlast || !dfeatures || l_hi >= m->layers || (!embed_only && (l_lo < 0 || l_lo > l_hi))) return MMT_ERR_ARG;
  Ws w;
  layout(m, b->rows_alloc, (char*)ws, &w);
  const int d = m->hidden, I = m->inter, rows = b->rows;
  const uint32_t th = training ? thr16_of(m->p_hidden) : 0u, ta = training ? thr16_of(m->p_attn) : 0u;
  const float sh = scale_of(th), sa = scale_of(ta);
  const int lh = b->n_rows_dev ? b->live_rows_hint : 0;  // the host's live row count (tile choice only: gemm.hip select_tile)
  const float qk_scale = m->hidden == m->heads * 128 ? 0.08838834764831845f : 0.125f;
  const int rpb = mmt_ln_bwd_rows_per_block(rows);
  const int ln_blocks = (rows + rpb - 1) / rpb;
  const int32_t* nr = b->n_rows_dev;

  void* side = (b->side_stream && b->side_stream != stream) ? b->side_stream : nullptr;
  const int fork = side ? b->fork : 0;
  const bool fork_w = fork & MMT_FORK_WGRAD, fork_early = fork_w && (fork & MMT_FORK_EARLY);
  const bool fork_r = fork & MMT_FORK_REDUCE, join = fork & MMT_FORK_JOIN;
  void* wstream = fork_w ? side : stream;   // weight gradients
  void* rstream = fork_r ? side : stream;   // LayerNorm / table reductions
  DoneEvents* done_e = fork_w ? done_entry(ws) : nullptr;
  if (fork_w && !done_e) return MMT_ERR_ARG;
  hipEvent_t* done = done_e ? done_e->ev : nullptr;
  unsigned long long* done_cap = done_e ? done_e->cap : nullptr;

  // LayerNorm gamma/beta and embedding-table partial sums stay in per-site buffers; ONE batched reduction at the end
  MmtColReduceJob jobs[2 * 64 + 5];
  int njobs = 0;
  auto add_job = [&](const float* partials, int nblocks, int nvec, int nout, int dd, float* o0, float* o1) {
    MmtColReduceJob& j = jobs[njobs++];
    j = {};
    j.partials = partials; j.nblocks = nblocks; j.nvec = nvec; j.nout = nout; j.d = dd; j.out[0] = o0; j.out[1] = o1;
  };
  auto finish = [&]() -> int {  // batched reduction (+ join) at the end of the range
    if (njobs) {
      if (fork_r) TRY(mmt_stream_fork(stream, side));
      TRY(mmt_col_reduce_multi(jobs, njobs, rstream));
    }
    if (join) TRY(mmt_stream_fork(side, stream));
    return 0;
  };
  const int nc = tail_rows(b, w);
  // r06: the backward's GEMM launches carry the caller's optimizer queue (MmtEpilogue.rider, include/mmt_hip.h "Adam
  // riders"): their idle blocks run the Adam update of parameters whose gradients are final -- everything the layers above
  // (and whatever ran before this call) have produced: rider_limits[l] entries while layer l's backward runs.  Launch j of
  // layer l reports its finished blocks in counter rider_slot0 + 8 l + j.
  int ride_j = 0;
  auto ride = [&](MmtEpilogue& e, int l) {
    if (!b->rider || !b->rider_limits || b->rider_limits[l] <= 0 || ride_j >= 8) return;
    const int slot = (b->rider_slot0 & 0xffff) + 8 * l + ride_j;
    if (slot < 0 || slot >= MMT_RIDER_SLOTS) return;
    e.rider = b->rider; e.rider_limit = b->rider_limits[l]; e.rider_slot = slot; e.rider_cap = (b->rider_slot0 >> 16) & 0xffff;
    ++ride_j;
  };
  // gradient wrt the current layer's output: ping-pongs between the caller's buffer and dA, starting at the top layer
  // (the buffer that holds the gradient wrt layer l's OUTPUT depends on l alone; "layer -1" = the embedding stage)
  float* dcur = ((m->layers - 1 - l_hi) & 1) ? w.dA : dlast;
  const float* dc_slabs = nullptr;  // != null: the gradient wrt the current layer's output is sum(slabs) + w.dz, not *dcur
  int dc_sp = 0;
  int64_t dc_stride = 0;
  for (int l = l_hi; l >= l_lo && !embed_only; --l) {
    const MmtBertLayer& P = m->layer[l];
    LayerWs& L = w.layer[l];
    const char* hin16 = l ? w.layer[l - 1].h16 : w.h16_in;
    const int par = fork_w ? (l & 1) : 0;  // one set unless weight gradients may still be running from two layers up
    ride_j = 0;
    char *dy = w.dy_[par], *dy2 = w.dy2_[par], *dhpre = w.dhpre_[par], *dqkv = w.dqkv_[par];
    // the weight gradients of layer l + 2 read the buffers this layer is about to overwrite (only an issue when they
    // were forked and not joined since: with MMT_FORK_JOIN every range call ends with the side stream drained)
    if (fork_w && !join && l + 2 <= m->layers - 1) {
      // under capture the event must have been recorded in THIS capture (ranges captured as separate graphs: MMT_FORK_JOIN)
      if (capture_id((hipStream_t)stream) != done_cap[par]) return MMT_ERR_ARG;
      if (hipStreamWaitEvent((hipStream_t)stream, done[par], 0) != hipSuccess) return MMT_ERR_ARG;
    }
    if (nc && l == m->layers - 1) {
      // ---- last layer on the nc read-out rows only (mirror of the forward tail) ----
      TailWs& t = w.t;
      const int crpb = mmt_ln_bwd_rows_per_block(nc);
      const int cblocks = (nc + crpb - 1) / crpb;
      // dlast holds the gradient of the COMPACT read-out rows in its first nc rows (it becomes scratch afterwards)
      TRY(mmt_ln_bwd(dlast, t.z2, t.mean2, t.rstd2, P.ln2_g, t.dz, t.dy2, w.ln_partials[2 * l + 2], nc, d, 1, nullptr,
                     t.rowidx, site_key(l, SITE_FFN_OUT), th, sh, b->seed_dev, stream));
      add_job(w.ln_partials[2 * l + 2], cblocks, 3, 2, d, P.g_ln2_g, P.g_ln2_b);
      MmtEpilogue e = {}; e.live_rows_hint = lh;
      e.aux = t.hpre; e.ldaux = I;
      ride(e, l);
      TRY(mmt_gemm_nt_bf16(t.dy2, d, P.w2_t, d, t.dhpre, I, nc, I, d, MMT_EPI_DGELU, &e, nullptr, stream));
      {  // dA = dhpre . W1 + dz: the split-K slabs and the residual gradient are summed by the LayerNorm backward itself
        int sp = 0;
        int64_t sstride = 0;
        TRY(mmt_gemm_splitk_geometry(nc, d, I, 0, &sp, &sstride));
        MmtEpilogue er = {}; er.live_rows_hint = lh;
        ride(er, l);
        TRY(mmt_gemm_nt_splitk_ex(t.dhpre, I, P.w1_t, I, nullptr, d, nc, d, I, MMT_EPI_F32, &er, t.slabs, 0, 0, nullptr, 1,
                                  stream));
        // t.dz is read (residual gradient) and rewritten (LN1 input gradient) by the same lanes at the same elements
        TRY(mmt_ln_bwd_slabs(t.slabs, sp, sstride, t.dz, t.z1, t.mean1, t.rstd1, P.ln1_g, t.dz, t.dy, w.ln_partials[2 * l + 1],
                             nc, d, 1, t.rowidx, site_key(l, SITE_ATTN_OUT), th, sh, b->seed_dev, stream));
      }
      add_job(w.ln_partials[2 * l + 1], cblocks, 3, 2, d, P.g_ln1_g, P.g_ln1_b);
      e = {}; e.live_rows_hint = lh;
      // (K = hidden: 8..16 K-steps -- one pass on 16+ tiles costs what the split-K slab kernel alone does, and the
      // slab-reducing epilogue launch goes away)
      // (the GEMM that forms dO also leaves rowsum(dO * O) per 64 columns = the attention backward's delta, in t.delta)
      e.dot_src = t.ctx; e.lddot = d; e.dot_out = t.delta;
      ride(e, l);
      if (d <= 512) TRY(mmt_gemm_nt_bf16(t.dy, d, P.wo_t, d, t.dctx, d, nc, d, d, MMT_EPI_BF16, &e, nullptr, stream));
      else TRY(mmt_gemm_nt_splitk(t.dy, d, P.wo_t, d, t.dctx, d, nc, d, d, MMT_EPI_BF16, &e, t.slabs, stream));
      // dQ exists for the read-out rows only (the dq kernel zero-fills the rest of the Q section); the residual
      // gradient t.dz likewise: the input-gradient GEMM runs without residual and t.dz is scatter-added afterwards
      TRY(mmt_attn_bwd_rows_ex(L.qkv, b->cu_seqlens, b->mask_bias, b->out_rows, b->n_out_per_sample, t.ctx, t.lse, t.dctx,
                               dqkv, t.delta, 1, b->batch, b->seq, m->heads, d, qk_scale, site_key(l, SITE_PROBS), ta, sa,
                               b->seed_dev, b->row_index, stream));
      if (fork_w) TRY(mmt_stream_fork(stream, side));  // every operand of the layer's weight gradients exists now
      {
        MmtWgradGroup g = {};
        g.count = 4; g.rows = rows; g.n_rows_dev = nr;
        g.item[0].A = dqkv;    g.item[0].lda = 3 * d; g.item[0].B = hin16; g.item[0].ldb = d; g.item[0].N = 3 * d; g.item[0].K2 = d;
        g.item[0].out = P.g_wqkv; g.item[0].bias_out = P.g_bqkv;
        g.item[0].splits = TAIL_WSPLIT; g.item[0].slab = t.wslab; g.item[0].bias_slab = t.bslab;
        g.item[1].A = t.dhpre; g.item[1].lda = I;     g.item[1].B = t.a16; g.item[1].ldb = d; g.item[1].N = I;     g.item[1].K2 = d;
        g.item[1].out = P.g_w1;   g.item[1].bias_out = P.g_b1; g.item[1].reserved = nc;
        g.item[2].A = t.dy2;   g.item[2].lda = d;     g.item[2].B = t.g;   g.item[2].ldb = I; g.item[2].N = d;     g.item[2].K2 = I;
        g.item[2].out = P.g_w2;   g.item[2].bias_out = P.g_b2; g.item[2].reserved = nc;
        g.item[3].A = t.dy;    g.item[3].lda = d;     g.item[3].B = t.ctx; g.item[3].ldb = d; g.item[3].N = d;     g.item[3].K2 = d;
        g.item[3].out = P.g_wo;   g.item[3].bias_out = P.g_bo; g.item[3].reserved = nc;
        TRY(mmt_wgrad_grouped(&g, wstream));
        TRY(mmt_reduce_slabs_pair(t.wslab, (int64_t)3 * d * d, P.g_wqkv, t.bslab, (int64_t)3 * d, P.g_bqkv, TAIL_WSPLIT, wstream));
        if (fork_w) { if (hipEventRecord(done[par], (hipStream_t)side) != hipSuccess) return MMT_ERR_ARG; done_cap[par] = capture_id((hipStream_t)side); }
      }
      e = {}; e.live_rows_hint = lh;
      ride(e, l);
      float* dnext = w.dA;
      TRY(gemm_hidden(w, rows, d, dqkv, 3 * d, P.wqkv_t, 3 * d, dnext, d, 3 * d, MMT_EPI_F32, &e, nr, stream));
      TRY(mmt_rows_scatter(t.dz, b->out_rows, nc, d, dnext, 1, stream));
      dcur = dnext;
      continue;
    }
    MmtWgradGroup gffn = {}, gatt = {};  // FFN pair (dW1, dW2) and attention pair (dWqkv, dWo) of the layer
    gffn.count = 2; gffn.rows = rows; gffn.n_rows_dev = nr;
    gffn.item[0].A = dhpre; gffn.item[0].lda = I;     gffn.item[0].B = L.a16; gffn.item[0].ldb = d; gffn.item[0].N = I;     gffn.item[0].K2 = d;
    gffn.item[0].out = P.g_w1;   gffn.item[0].bias_out = P.g_b1;
    gffn.item[1].A = dy2;   gffn.item[1].lda = d;     gffn.item[1].B = L.g;   gffn.item[1].ldb = I; gffn.item[1].N = d;     gffn.item[1].K2 = I;
    gffn.item[1].out = P.g_w2;   gffn.item[1].bias_out = P.g_b2;
    gatt.count = 2; gatt.rows = rows; gatt.n_rows_dev = nr;
    gatt.item[0].A = dqkv;  gatt.item[0].lda = 3 * d; gatt.item[0].B = hin16; gatt.item[0].ldb = d; gatt.item[0].N = 3 * d; gatt.item[0].K2 = d;
    gatt.item[0].out = P.g_wqkv; gatt.item[0].bias_out = P.g_bqkv;
    gatt.item[1].A = dy;    gatt.item[1].lda = d;     gatt.item[1].B = L.ctx; gatt.item[1].ldb = d; gatt.item[1].N = d;     gatt.item[1].K2 = d;
    gatt.item[1].out = P.g_wo;   gatt.item[1].bias_out = P.g_bo;
    // --- BertOutput: LN2 <- dropout <- dense(I->d) ---
    if (dc_slabs) {  // the layer above left its input gradient as split-K slabs + the residual gradient in w.dz (see below)
      TRY(mmt_ln_bwd_slabs_ex(dc_slabs, dc_sp, dc_stride, w.dz, L.z2, L.mean2, L.rstd2, P.ln2_g, w.dz, dy2, w.ln_partials[2 * l + 2],
                              rows, d, 1, nr, b->row_index, site_key(l, SITE_FFN_OUT), th, sh, b->seed_dev, stream));
      dc_slabs = nullptr;
    } else {
    TRY(mmt_ln_bwd(dcur, L.z2, L.mean2, L.rstd2, P.ln2_g, w.dz, dy2, w.ln_partials[2 * l + 2], rows, d, 1, nr, b->row_index,
                   site_key(l, SITE_FFN_OUT), th, sh, b->seed_dev, stream));
    }
    add_job(w.ln_partials[2 * l + 2], ln_blocks, 3, 2, d, P.g_ln2_g, P.g_ln2_b);
    MmtEpilogue e = {}; e.live_rows_hint = lh;
    e.aux = L.hpre; e.ldaux = I;
    ride(e, l);
    TRY(mmt_gemm_nt_bf16(dy2, d, P.w2_t, d, dhpre, I, rows, I, d, MMT_EPI_DGELU, &e, nr, stream));
    if (fork_early) {  // dW1 / dW2 need nothing else: they start under the rest of this layer's input-gradient chain
      TRY(mmt_stream_fork(stream, side));
      TRY(mmt_wgrad_grouped(&gffn, side));
    }
    // --- BertIntermediate: dense(d->I) ---
    const int small_sp = small_splitk_ln(w, rows, d, I);
    if (splitk_ffn(w, rows, d, I) || small_sp) {
      const int mode = small_sp ? 10 * small_sp : splitk_ffn_mode();
      float* slabs = small_sp ? w.t.slabs : w.kslab;
      int sp = 0;
      int64_t sstride = 0;
      TRY(mmt_gemm_splitk_geometry(rows, d, I, mode / 10, &sp, &sstride));
      MmtEpilogue er = {}; er.live_rows_hint = lh;
      ride(er, l);
      TRY(mmt_gemm_nt_splitk_ex(dhpre, I, P.w1_t, I, nullptr, d, rows, d, I, MMT_EPI_F32, &er, slabs, mode / 10, mode % 10,
                                nr, 1, stream));
      // w.dz is read (residual gradient) and rewritten (LN1 input gradient) by the same lanes at the same elements
      TRY(mmt_ln_bwd_slabs_ex(slabs, sp, sstride, w.dz, L.z1, L.mean1, L.rstd1, P.ln1_g, w.dz, dy, w.ln_partials[2 * l + 1],
                              rows, d, 1, nr, b->row_index, site_key(l, SITE_ATTN_OUT), th, sh, b->seed_dev, stream));
    } else {
    e = {}; e.live_rows_hint = lh;
    e.res = w.dz; e.ldres = d;
    ride(e, l);
    TRY(gemm_hidden(w, rows, d, dhpre, I, P.w1_t, I, w.dA, d, I, MMT_EPI_ADD_F32, &e, nr, stream));
    // --- BertSelfOutput: LN1 <- dropout <- dense(d->d) ---
    TRY(mmt_ln_bwd(w.dA, L.z1, L.mean1, L.rstd1, P.ln1_g, w.dz, dy, w.ln_partials[2 * l + 1], rows, d, 1, nr, b->row_index,
                   site_key(l, SITE_ATTN_OUT), th, sh, b->seed_dev, stream));
    }
    add_job(w.ln_partials[2 * l + 1], ln_blocks, 3, 2, d, P.g_ln1_g, P.g_ln1_b);
    e = {}; e.live_rows_hint = lh;
    e.dot_src = L.ctx; e.lddot = d; e.dot_out = w.delta;  // rowsum(dO * O) per 64 columns, while dO is in registers
    ride(e, l);
    TRY(mmt_gemm_nt_bf16(dy, d, P.wo_t, d, w.dctx, d, rows, d, d, MMT_EPI_BF16, &e, nr, stream));
    // --- BertSelfAttention ---
    {
      ProbeScope probe(4, l == 0, stream);
      TRY(mmt_attn_bwd_ex(L.qkv, b->cu_seqlens, b->mask_bias, L.ctx, L.lse, w.dctx, dqkv, w.delta, 1, b->batch, b->seq,
                          m->heads, d, qk_scale, site_key(l, SITE_PROBS), ta, sa, b->seed_dev, b->row_index, attn_work_of(m, b, w),
                          stream));
    }
    if (fork_w) {
      // --- weight + bias gradients on the side stream, under the input-gradient GEMM below and the layers that follow ---
      TRY(mmt_stream_fork(stream, side));
      if (fork_early) {
        TRY(mmt_wgrad_grouped(&gatt, side));
      } else {
        MmtWgradGroup g = gffn;
        g.count = 4; g.item[2] = gatt.item[0]; g.item[3] = gatt.item[1];
        ProbeScope probe(2, l == 0, side);
        TRY(mmt_wgrad_grouped(&g, side));
      }
      if (hipEventRecord(done[par], (hipStream_t)side) != hipSuccess) return MMT_ERR_ARG;
      done_cap[par] = capture_id((hipStream_t)side);
    }
    e = {}; e.live_rows_hint = lh;
    e.res = w.dz; e.ldres = d;
    float* dnext = (dcur == dlast) ? w.dA : dlast;  // ping-pong between the caller's buffer and dA
    // dA was consumed by the LN1 backward above, so it is free again here.
    ride(e, l);
    // r06, short batches: the split-K slabs of dX = dQKV . Wqkv are summed (+ the residual gradient w.dz) by the LayerNorm
    // backward of the layer below instead of by a slab-reducing epilogue launch -- when that layer runs in this call and
    // nothing is forked (the weight gradients below read dqkv, not the slabs; w.dz is not written in between)
    const int qkv_sp = (l > l_lo && !fork_w) ? small_splitk_ln(w, rows, d, 3 * d) : 0;
    if (qkv_sp) {
      TRY(mmt_gemm_splitk_geometry(rows, d, 3 * d, qkv_sp, &dc_sp, &dc_stride));
      TRY(mmt_gemm_nt_splitk_ex(dqkv, 3 * d, P.wqkv_t, 3 * d, nullptr, d, rows, d, 3 * d, MMT_EPI_F32, &e, w.t.slabs, qkv_sp, 0, nr, 1,
                                stream));
      dc_slabs = w.t.slabs;
    } else
    TRY(gemm_hidden(w, rows, d, dqkv, 3 * d, P.wqkv_t, 3 * d, dnext, d, 3 * d, MMT_EPI_ADD_F32, &e, nr, stream));
    // --- all four weight gradients + bias gradients of the layer: ONE grouped launch (256 tiles at d=512, I=3072) ---
    if (!fork_w) {
      MmtWgradGroup g = gffn;
      g.count = 4; g.item[2] = gatt.item[0]; g.item[3] = gatt.item[1];
      ProbeScope probe(2, l == 0, stream);
      TRY(mmt_wgrad_grouped(&g, stream));
    }
    dcur = dnext;
  }
  if (!embed_only && (l_lo > 0 || (b->fork & MMT_RANGE_LAYERS_ONLY))) return finish();
  // --- BertEmbeddings: dropout <- LN <- (features + type_emb + pos_emb) ---
  TRY(mmt_ln_bwd(dcur, w.z0, w.mean0, w.rstd0, m->emb_ln_g, dfeatures, nullptr, w.ln_partials[0], rows, d, 2, nr,
                 b->row_index, site_key(0, SITE_EMB), th, sh, b->seed_dev, stream));
  add_job(w.ln_partials[0], ln_blocks, 3, 2, d, m->g_emb_ln_g, m->g_emb_ln_b);
  if (fork_r) TRY(mmt_stream_fork(stream, side));
  const int chunks = mmt_table_grad_chunks();
  const bool pos_partials = b->pos_ids && m->max_pos <= 128;  // (one-hot MFMA products up to 4 vocabulary tiles of 32)
  // token-type table (+ the temporal-position table when it is small) as one-hot MFMA products, ONE launch for both
  if (m->type_vocab <= 128) {
    TRY(mmt_table_grad_partials_pair(dfeatures, b->type_ids, m->type_vocab, w.table_scratch[0],
                                     pos_partials ? b->pos_ids : nullptr, m->max_pos, w.table_scratch[1], rows, d, nr, rstream));
  } else {  // large token-type vocabularies: the scan kernel
    TRY(mmt_table_grad_partials(dfeatures, b->type_ids, rows, d, m->type_vocab, nr, w.table_scratch[0], rstream));
    if (pos_partials) TRY(mmt_table_grad_partials(dfeatures, b->pos_ids, rows, d, m->max_pos, nr, w.table_scratch[1], rstream));
  }
  add_job(w.table_scratch[0], chunks, 1, 1, m->type_vocab * d, m->g_type_emb, nullptr);
  if (pos_partials) add_job(w.table_scratch[1], chunks, 1, 1, m->max_pos * d, m->g_pos_emb, nullptr);
  else if (b->pos_ids)  // BERT-base position table (512 rows, a few dozen in use): no vocab-sized partial sums
    TRY(mmt_table_grad_direct(dfeatures, b->pos_ids, rows, d, m->max_pos, nr, m->g_pos_emb, rstream));
  // (the fork for the reductions happened above: `rstream` already sees the LayerNorm partials of this range)
  TRY(mmt_col_reduce_multi(jobs, njobs, rstream));
  if (join) TRY(mmt_stream_fork(side, stream));
  return 0;
}

extern "C" int mmt_bert_backward(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* dlast,
                                 float* dfeatures, int training, void* stream) {
  if (!m) return MMT_ERR_ARG;
  return mmt_bert_backward_range(m, b, ws, dlast, dfeatures, training, m->layers - 1, 0, stream);
}
