// qvc_api.hip -- the C ABI of include/qvc.h: argument checks + the launch sequence of the path.
//
// Everything here is host code that only *enqueues* kernels on the caller's stream: no
// allocation, no synchronisation, no host read of device data, so a whole qvc_infer_batch
// call can be captured into a hipGraph by the caller.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "qvc_path.h"
#include "qvc_stream.h"
#include "qvc_pack_util.h"

namespace qvc {
int launch_fm_to_cm(const float* src, float* dst, int batch, int frames, int channels, void* stream);
#ifdef QVC_SATCOUNT
unsigned long long sat_count_conv_f16(bool reset);
unsigned long long sat_count_conv_bf16(bool reset);
unsigned long long sat_count_wn2(bool reset);
unsigned long long sat_count_spk(bool reset);
unsigned long long sat_count_chain(bool reset);
#endif
}

struct qvc_aux {
  hipStream_t streams[2];
  hipEvent_t fork, done[3];
};

namespace {
using namespace qvc;

// Layers per whole-stack WaveNet launch.  The debug switch "wn_chunk" overrides it (tuning runs; -1 = no stack kernel
// at all: one fused launch per layer and the coupling layers' pre / post as separate convs -- the path taken by
// configurations the stack kernel does not cover, selectable so that the tests can exercise it).
inline int wn_chunk(int layers) {
  const int env = debug_get(DBG_WN_CHUNK);
  if (env < 0) return 0;
  if (env > 0) return env <= layers ? env : layers;
  return layers % 4 == 0 ? 4 : layers;
}

// Stream fork/join for the parallel ResBlock branches (branch 0 = the caller's stream).
struct Branches {
  hipStream_t main = nullptr;
  qvc_aux* aux = nullptr;
  hipStream_t cur = nullptr;
  bool ok = true;
  hipStream_t of(int j) const { return (aux && j >= 1 && j <= 2) ? aux->streams[j - 1] : main; }
  void fork(int n) {
    cur = main;
    if (!aux) return;
    if (hipEventRecord(aux->fork, main) != hipSuccess) ok = false;
    for (int j = 1; j < n && j <= 2; ++j) if (hipStreamWaitEvent(aux->streams[j - 1], aux->fork, 0) != hipSuccess) ok = false;
  }
  void branch(int j) { cur = of(j); }
  void branch_done(int j) { if (aux && j < 3 && hipEventRecord(aux->done[j], of(j)) != hipSuccess) ok = false; }
  void wait_branch_done(int j) { if (aux && j < 3 && hipStreamWaitEvent(cur, aux->done[j], 0) != hipSuccess) ok = false; }
  void join(int n) {
    if (aux)     // branches 1 and 2 ran on the auxiliary streams (further ones, and branch 0, on main)
      for (int j = 1; j < n && j <= 2; ++j)
        if (hipStreamWaitEvent(main, aux->done[j], 0) != hipSuccess) ok = false;
    cur = main;
  }
};

struct HipBackend {
  hipStream_t stream;
  Branches br;
  void fork(int n) { br.main = stream0; br.fork(n); stream = br.cur; }
  void branch(int j) { br.branch(j); stream = br.cur; }
  void branch_done(int j) { br.branch_done(j); }
  void wait_branch_done(int j) { br.wait_branch_done(j); }
  void join(int n) { br.join(n); stream = stream0; }
  hipStream_t stream0 = nullptr;
  int conv(const ConvDesc& d, const ConvArgs& a, int batch, int epi, int dtype) { return launch_conv(d, a, batch, epi, dtype, stream); }
  int pair(const ConvDesc& d1, const ConvDesc& d2, const PairArgs& a, int batch, int dtype) { return launch_pair(d1, d2, a, batch, dtype, stream); }
  int pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3& a, int batch, int dtype) { return launch_pair3(d1, d2, a, batch, dtype, stream); }
  bool chain_ok(const ConvDesc* d1, const ConvDesc* d2, int n) const { return debug_get(DBG_PAIR_CHAIN3) != 0 && chain_supported(d1, d2, n); }
  int chain(const ConvDesc* d1, const ConvDesc* d2, const ChainArgs& a, int batch, int dtype) { return launch_chain(d1, d2, a, batch, dtype, stream); }
  int wn(const ConvDesc& din, const ConvDesc&, const WnArgs& a, int batch, int dtype) { return launch_wn(din, a, batch, dtype, stream); }
  // the stack kernel recomputes halo frames but a layer is bound by its weight stream, not by MFMA work: measured
  // no slower than one launch per layer at any batch (16.4 vs 17.9 us per layer at batch 1)
  bool use_wn_stack(int, int) const { return true; }
  int wn_stack_chunk(int layers) const { return wn_chunk(layers); }
  int wn_stack(const ConvDesc& din, const ConvDesc&, const ConvDesc&, const WnStackArgs& a, int batch, int dtype, const ConvDesc*, const ConvDesc*) { return launch_wn_stack(din, a, batch, dtype, stream); }
  int gemv(const GemvArgs& a) { return launch_gemv(a, stream); }
  int sample(const SampleArgs& a) { return launch_sample(a, stream); }
  int tail(const TailArgs& a) { return launch_tail(a, stream); }
  bool post_tail_ok(const ConvDesc& d) const { return debug_get(DBG_POST_TAIL) != 0 && post_tail_supported(d); }
  int post_tail(const ConvDesc& d, const PostTailArgs& a, int batch, int dtype) { return launch_post_tail(d, a, batch, dtype, stream); }
  int zero(void* p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, stream) == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH; }
  int copy_batch(const CopyDesc* d, int n) { return launch_copy_batch(d, n, stream); }
};
using Ctx = Path<HipBackend>;

// Same launches, each bracketed by events on the stream (diagnostics only).
struct TimedBackend {
  void fork(int) {} void branch(int) {} void branch_done(int) {} void wait_branch_done(int) {} void join(int) {}
  hipStream_t stream;
  qvc_launch_record* rec; int max_rec; int n = 0;
  std::vector<hipEvent_t> ev;
  bool ok = true;
  void mark() { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, stream) != hipSuccess) ok = false; ev.push_back(e); }
  void note(const char* name, double flops, double bytes) {
    if (n < max_rec) { std::snprintf(rec[n].name, sizeof(rec[n].name), "%s", name); rec[n].flops = flops; rec[n].bytes = bytes; rec[n].ms = 0.f; }
    ++n;
  }
  int conv(const ConvDesc& d, const ConvArgs& a, int batch, int epi, int dtype) {
    if (ev.empty()) mark();
    int nf = 0;
    int st = launch_conv(d, a, batch, epi, dtype, stream, &nf);
    mark();
    char name[48];
    std::snprintf(name, sizeof(name), "conv<%s,MF%d,NF%d,WM%d,%s>", dtype == QVC_F16 ? "f16" : "bf16", d.MF, nf, d.WM, epi == EPI_GAU ? "gau" : (epi == EPI_SAMPLE ? "smp" : "std"));
    // algorithmic work: a transposed conv does k MACs per (input frame, ci, co), a conv taps MACs per output
    const double macs = d.up_s > 1 ? (double)batch * a.T_in * d.Cin * d.Cout * (double)d.ksize
                                   : (double)batch * a.Nq * (double)d.M * d.taps * d.Cin;
    const double in_b = (double)batch * a.T_in * d.Cin * (a.x_kind == XK_OP_FM ? 2 : 4) * (a.x2 ? 3 : 1);
    double out_b = 0;
    const double outs = (double)batch * a.T_out * d.Cout;
    if (epi == EPI_GAU) out_b = (double)batch * a.Nq * a.gau_H * 2;
    else if (epi == EPI_SAMPLE) out_b = (double)batch * a.Nq * a.gau_H * 8;      // noise in, z out
    else {
      if (a.y32) out_b += outs * 4 * (a.y_accum ? 2 : 1);
      if (a.y16) out_b += outs * 2;
      if (a.res) out_b += outs * 4;
      if (a.res16) out_b += outs * 2;
      if (a.y32b) out_b += (double)batch * a.Nq * (d.M - a.split) * 8;
    }
    note(name, 2.0 * macs, in_b + out_b + (double)d.w_bytes());
    return st;
  }
  int pair(const ConvDesc& d1, const ConvDesc& d2, const PairArgs& a, int batch, int dtype) {
    PairArgs3 a3; a3.p[0] = a; a3.n = 1;
    return pair3(&d1, &d2, a3, batch, dtype);
  }
  int pair3(const ConvDesc* d1, const ConvDesc* d2, const PairArgs3& a, int batch, int dtype) {
    if (ev.empty()) mark();
    int nf = 0;
    int st = launch_pair3(d1, d2, a, batch, dtype, stream, &nf);
    mark();
    char name[48];
    // one kernel symbol serves 1..3 chains per launch; the record's flops / bytes are those of the whole launch
    const char* tn = dtype == QVC_F16 ? "f16" : (dtype == QVC_BF16X ? "bf16x" : "bf16");
    if (nf >= 100) std::snprintf(name, sizeof(name), "rbpair_persist<%s,MF%d,NF%d>", tn, d1[0].MF, nf - 100);
    else std::snprintf(name, sizeof(name), "rbpair<%s,MF%d,NF%d,WM%d>", tn, d1[0].MF, nf, d1[0].WM);
    double fl = 0, by = 0;
    for (int i = 0; i < a.n; ++i) {
      const double outs = (double)batch * a.p[i].T * a.p[i].C;
      fl += 2.0 * 2.0 * outs * a.p[i].C * a.p[i].k;
      by += outs * 2 * 2 + (double)d1[i].w_bytes() + (double)d2[i].w_bytes();   // algorithmic: the stream read once, written once (the kernel's second read of x for the residual is NOT counted: it shows up as traffic / algorithmic > 1)
    }
    note(name, fl, by);
    return st;
  }
  bool chain_ok(const ConvDesc* d1, const ConvDesc* d2, int n) const { return debug_get(DBG_PAIR_CHAIN3) != 0 && chain_supported(d1, d2, n); }
  int chain(const ConvDesc* d1, const ConvDesc* d2, const ChainArgs& a, int batch, int dtype) {
    if (ev.empty()) mark();
    int nf = 0;
    int st = launch_chain(d1, d2, a, batch, dtype, stream, &nf);
    mark();
    char name[48];
    const char* tn = dtype == QVC_F16 ? "f16" : (dtype == QVC_BF16X ? "bf16x" : "bf16");
    std::snprintf(name, sizeof(name), "rbchain<%s,MF%d,NF%d,WM%d,k%d>", tn, d1[0].MF, nf, d1[0].WM, d1[0].taps);
    const double outs = (double)batch * a.p[0].T * a.p[0].C;
    note(name, a.n * 2.0 * 2.0 * outs * a.p[0].C * a.p[0].k, outs * 2 * 2 + a.n * ((double)d1[0].w_bytes() + (double)d2[0].w_bytes()));
    return st;
  }
  int wn(const ConvDesc& din, const ConvDesc& drs, const WnArgs& a, int batch, int dtype) {
    if (ev.empty()) mark();
    int nf = 0;
    int st = launch_wn(din, a, batch, dtype, stream, &nf);
    mark();
    char name[48];
    std::snprintf(name, sizeof(name), "wn_layer<%s,W%d,NF%d%s>", dtype == QVC_F16 ? "f16" : "bf16", din.WM, nf, a.last ? ",last" : "");
    const double cols = (double)batch * a.T;
    note(name, 2.0 * cols * a.H * (2.0 * a.H * a.taps + (double)drs.M), cols * a.H * 4 * (a.last ? 3 : 5) + (double)din.w_bytes() + (double)drs.w_bytes());
    return st;
  }
  bool use_wn_stack(int, int) const { return true; }
  int wn_stack_chunk(int layers) const { return wn_chunk(layers); }
  int wn_stack(const ConvDesc& din, const ConvDesc& drs, const ConvDesc& drs_last, const WnStackArgs& a, int batch, int dtype,
               const ConvDesc* dpre, const ConvDesc* dpost) {
    if (ev.empty()) mark();
    int st = launch_wn_stack(din, a, batch, dtype, stream);
    mark();
    char name[48];
    std::snprintf(name, sizeof(name), "%s<%s,W%d,L%d%s>", wn_stack_variant(din, a) == 2 ? "wn_stack2" : "wn_stack", dtype == QVC_F16 ? "f16" : "bf16",
                  din.WM, a.layers, dpre ? ",pre+post" : "");
    const double cols = (double)batch * a.T;
    const double fl = 2.0 * cols * a.H * (a.layers * 2.0 * a.H * a.taps + (a.layers - (a.final_layer ? 1 : 0)) * (double)drs.M + (a.final_layer ? (double)drs_last.M : 0.0));
    const double fuse_fl = (dpre ? 2.0 * cols * dpre->M * dpre->Cin : 0.0) + (dpost ? 2.0 * cols * dpost->M * dpost->Cin : 0.0);
    note(name, fl + fuse_fl, cols * a.H * 8 + a.layers * ((double)din.w_bytes() + (double)drs.w_bytes()));
    return st;
  }
  int gemv(const GemvArgs& a) { if (ev.empty()) mark(); int st = launch_gemv(a, stream); mark();
    note("cond_gemv", 2.0 * a.rows * a.gin * a.batch, (double)a.rows * a.gin * 4 + (double)a.batch * (a.rows + a.gin) * 4); return st; }
  int sample(const SampleArgs& a) { if (ev.empty()) mark(); int st = launch_sample(a, stream); mark();
    note("sample", 0, (double)a.batch * a.frames * a.C * 16); return st; }
  int tail(const TailArgs& a) { if (ev.empty()) mark(); int st = launch_tail(a, stream); mark();
    note("istft_synth", 0, (double)a.batch * ((double)a.F * 72 * 4 + 16.0 * (a.F - 1) * 4)); return st; }
  bool post_tail_ok(const ConvDesc& d) const { return debug_get(DBG_POST_TAIL) != 0 && post_tail_supported(d); }
  int post_tail(const ConvDesc& d, const PostTailArgs& a, int batch, int dtype) {
    if (ev.empty()) mark();
    int st = launch_post_tail(d, a, batch, dtype, stream);
    mark();
    char name[48];
    std::snprintf(name, sizeof(name), "post_tail<%s>", dtype == QVC_F16 ? "f16" : "bf16");
    note(name, 2.0 * batch * (double)a.F * d.M * d.taps * d.Cin,
         (double)batch * ((double)a.c.T_in * d.Cin * 2 * 3 + 16.0 * (a.F - 1) * 4) + (double)d.w_bytes());
    return st;
  }
  int zero(void* p, size_t bytes) { if (ev.empty()) mark(); int st = hipMemsetAsync(p, 0, bytes, stream) == hipSuccess ? QVC_OK : QVC_ERR_LAUNCH; mark();
    note("memset", 0, (double)bytes); return st; }
  int finish() {
    if (!ok || hipStreamSynchronize(stream) != hipSuccess) ok = false;
    for (int i = 0; i + 1 < (int)ev.size() && i < max_rec; ++i) {
      float ms = 0.f;
      if (ok && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) rec[i].ms = ms;
    }
    for (hipEvent_t e : ev) hipEventDestroy(e);
    ev.clear();
    return ok ? QVC_OK : QVC_ERR_LAUNCH;
  }
};

int check_common(const qvc_config* cfg, const void* blob, int batch, int frames, void* ws, int64_t ws_bytes, Plan& P,
                 Workspace& W) {
  if (!cfg || !blob || !ws || batch <= 0 || frames <= 1) return QVC_ERR_BAD_ARG;
  P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  W = carve_workspace(P, batch, frames);
  if (ws_bytes < W.bytes) return QVC_ERR_SMALL_BUFFER;
  if ((reinterpret_cast<uintptr_t>(ws) & 255) || (reinterpret_cast<uintptr_t>(blob) & 255)) return QVC_ERR_BAD_ARG;
  return QVC_OK;
}

// qvc_infer_batch_ragged / _fm: the whole path over utterances of different lengths (unit_fm: units frame-major as on disk)
int infer_ragged(const qvc_config* cfg, const void* blob_dev, const float* unit, bool unit_fm, const float* g, const float* noise,
                 float* out, int32_t batch, int32_t max_frames, const int32_t* frames_dev, void* workspace, int64_t workspace_bytes,
                 void* stream) {
  if (!unit || !g || !noise || !out || !frames_dev) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, max_frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, max_frames, be};
  c.lens = frames_dev;
  c.unit_fm = unit_fm;
  c.cond_table(g);
  c.enc_p(unit, noise, c.wsp<float>(W.z));
  c.flow(c.wsp<float>(W.z));
  c.dec_trunk_wave(c.wsp<float>(W.z), c.wsp<float>(W.post), out);
  return c.status != QVC_OK ? c.status : (be.br.ok ? QVC_OK : QVC_ERR_LAUNCH);
}

}  // namespace

extern "C" {

int qvc_abi_version(void) { return QVC_ABI_VERSION; }

const char* qvc_status_string(int status) {
  switch (status) {
    case QVC_OK: return "ok";
    case QVC_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size, misaligned buffer or bad enum)";
    case QVC_ERR_BAD_CONFIG: return "unsupported model configuration";
    case QVC_ERR_MISSING_TENSOR: return "a state-dict tensor needed by the path is missing";
    case QVC_ERR_BAD_SHAPE: return "a state-dict tensor has an unexpected shape";
    case QVC_ERR_SMALL_BUFFER: return "blob or workspace smaller than the matching *_bytes() query";
    case QVC_ERR_LAUNCH: return "HIP kernel launch failed";
    case QVC_ERR_NO_DEVICE: return "no gfx950 HIP device visible";
    default: return "unknown status";
  }
}

int qvc_debug_set(const char* name, int32_t value) {
  if (!name) return QVC_ERR_BAD_ARG;
  for (int i = 0; i < DBG_COUNT; ++i)
    if (std::strcmp(name, debug_names()[i]) == 0) { debug_table()[i].store(value, std::memory_order_relaxed); return QVC_OK; }
  return QVC_ERR_BAD_ARG;
}

int qvc_debug_get(const char* name, int32_t* value) {
  if (!name || !value) return QVC_ERR_BAD_ARG;
  for (int i = 0; i < DBG_COUNT; ++i)
    if (std::strcmp(name, debug_names()[i]) == 0) { *value = debug_get(i); return QVC_OK; }
  return QVC_ERR_BAD_ARG;
}

int qvc_debug_saturations(int64_t* count, int32_t reset) {
  if (!count) return QVC_ERR_BAD_ARG;
#ifdef QVC_SATCOUNT
  const unsigned long long v[5] = {sat_count_conv_f16(reset != 0), sat_count_conv_bf16(reset != 0), sat_count_wn2(reset != 0), sat_count_spk(reset != 0),
                                   sat_count_chain(reset != 0)};
  unsigned long long sum = 0;
  for (unsigned long long x : v) { if (x == ~0ull) return QVC_ERR_LAUNCH; sum += x; }
  *count = (int64_t)sum;
  return QVC_OK;
#else
  (void)reset;
  *count = -1;
  return QVC_ERR_BAD_CONFIG;      // the product library does not count (no instruction is spent on it)
#endif
}

int qvc_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return QVC_ERR_NO_DEVICE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return QVC_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return QVC_ERR_NO_DEVICE;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? QVC_OK : QVC_ERR_NO_DEVICE;
}

int64_t qvc_workspace_bytes(const qvc_config* cfg, int32_t batch, int32_t frames) {
  if (!cfg || batch <= 0 || frames <= 1) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  return carve_workspace(P, batch, frames).bytes;
}

int qvc_plan_info(const qvc_config* cfg, int32_t info[8]) {
  if (!cfg || !info) return QVC_ERR_BAD_ARG;
  const Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  for (int i = 0; i < 8; ++i) info[i] = 0;
  info[0] = P.enc_proj.gau; info[1] = P.enc_proj.MF;
  for (size_t i = 0; i < P.stages.size() && i < 2; ++i) {
    info[2 + i] = P.stages[i].up.lp;
    info[5 + i] = block_waves(P.stages[i].c1[0]);
  }
  info[4] = post_tail_supported(P.conv_post) ? 1 : 0;
  int all = P.cfg.n_resblocks <= 3 ? 1 : 0;
  for (const StagePlan& st : P.stages)
    for (size_t j = 0; j < st.c1.size(); ++j)
      all = all && pair_supported(st.c1[j], st.c2[j]) && st.c1[j].lp && st.c2[j].lp && st.c1[j].MF == st.c1[0].MF && st.c1[j].WM == st.c1[0].WM;
  info[7] = all;
  return QVC_OK;
}

int qvc_aux_create(qvc_aux** out) {
  if (!out) return QVC_ERR_BAD_ARG;
  qvc_aux* a = new (std::nothrow) qvc_aux();
  if (!a) return QVC_ERR_BAD_ARG;
  bool ok = true;
  for (auto& s : a->streams) ok = ok && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&a->fork, hipEventDisableTiming) == hipSuccess;
  for (auto& e : a->done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  if (!ok) { delete a; return QVC_ERR_LAUNCH; }
  *out = a;
  return QVC_OK;
}

int qvc_aux_destroy(qvc_aux* a) {
  if (!a) return QVC_ERR_BAD_ARG;
  for (auto& s : a->streams) hipStreamDestroy(s);
  hipEventDestroy(a->fork);
  for (auto& e : a->done) hipEventDestroy(e);
  delete a;
  return QVC_OK;
}

int qvc_infer_batch(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* g,
                    const float* noise, float* out, int32_t batch, int32_t frames, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  return qvc_infer_batch_ex(cfg, blob_dev, unit, g, noise, out, batch, frames, workspace, workspace_bytes, stream, nullptr);
}

int qvc_infer_batch_ex(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* g,
                       const float* noise, float* out, int32_t batch, int32_t frames, void* workspace,
                       int64_t workspace_bytes, void* stream, qvc_aux* aux) {
  if (!unit || !g || !noise || !out) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream); be.br.aux = aux;
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.cond_table(g);
  c.enc_p(unit, noise, c.wsp<float>(W.z));
  c.flow(c.wsp<float>(W.z));
  c.dec_trunk_wave(c.wsp<float>(W.z), c.wsp<float>(W.post), out);
  return c.status != QVC_OK ? c.status : (be.br.ok ? QVC_OK : QVC_ERR_LAUNCH);
}

int qvc_infer_batch_ragged(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* g,
                           const float* noise, float* out, int32_t batch, int32_t max_frames, const int32_t* frames_dev,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  return infer_ragged(cfg, blob_dev, unit, false, g, noise, out, batch, max_frames, frames_dev, workspace, workspace_bytes, stream);
}

int qvc_infer_batch_ragged_fm(const qvc_config* cfg, const void* blob_dev, const float* unit_fm, const float* g,
                              const float* noise, float* out, int32_t batch, int32_t max_frames, const int32_t* frames_dev,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  return infer_ragged(cfg, blob_dev, unit_fm, true, g, noise, out, batch, max_frames, frames_dev, workspace, workspace_bytes, stream);
}

int64_t qvc_stream_state_bytes(const qvc_config* cfg, int32_t batch, int32_t hop) {
  if (!cfg || batch <= 0 || hop <= 0) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  return carve_stream_state(P, G, batch).bytes;
}

int64_t qvc_stream_workspace_bytes(const qvc_config* cfg, int32_t batch, int32_t hop) {
  if (!cfg || batch <= 0 || hop <= 0) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  return carve_stream_scratch(P, G, batch).bytes;
}

int32_t qvc_stream_lag_frames(const qvc_config* cfg) {
  if (!cfg) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, 1);
  return G.status != QVC_OK ? G.status : G.lag();
}

int32_t qvc_stream_noise_lag_frames(const qvc_config* cfg) {
  if (!cfg) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, 1);
  return G.status != QVC_OK ? G.status : G.He;
}

int qvc_stream_step(const qvc_config* cfg, const void* blob_dev, void* state, int64_t state_bytes, const float* unit_new,
                    const float* g, const float* noise_new, float* out, int32_t batch, int32_t hop, const int32_t* pos_dev,
                    const int32_t* len_dev, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!cfg || !blob_dev || !state || !unit_new || !g || !noise_new || !out || !pos_dev || !len_dev || !workspace || batch <= 0 || hop <= 0)
    return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  if (state_bytes < carve_stream_state(P, G, batch).bytes || workspace_bytes < carve_stream_scratch(P, G, batch).bytes) return QVC_ERR_SMALL_BUFFER;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(blob_dev) & 255) || (reinterpret_cast<uintptr_t>(state) & 255))
    return QVC_ERR_BAD_ARG;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  const int st = stream_step(P, static_cast<const char*>(blob_dev), static_cast<char*>(state), static_cast<char*>(workspace), unit_new, g,
                             noise_new, out, batch, hop, pos_dev, len_dev, be);
  return st != QVC_OK ? st : (be.br.ok ? QVC_OK : QVC_ERR_LAUNCH);
}

int qvc_stream_reset_slot(const qvc_config* cfg, void* state, int64_t state_bytes, int32_t batch, int32_t hop, int32_t slot,
                          int32_t length, int32_t* pos_dev, int32_t* len_dev, void* stream) {
  if (!cfg || !state || !pos_dev || !len_dev || batch <= 0 || hop <= 0 || slot < 0 || slot >= batch || length < 0) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  const StreamGeom G = stream_geom(P, hop);
  if (G.status != QVC_OK) return G.status;
  const StreamState S = carve_stream_state(P, G, batch);
  if (state_bytes < S.bytes) return QVC_ERR_SMALL_BUFFER;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(state);
  const qvc_config& c = P.cfg;
  bool ok = true;
  // every ring is batch-major, so a stream's rows are one contiguous range per ring
  auto zero = [&](int64_t ring, int64_t per_stream) {
    ok = ok && hipMemsetAsync(base + ring + (int64_t)slot * per_stream, 0, (size_t)per_stream, st) == hipSuccess;
  };
  zero(S.unit, (int64_t)c.unit_channels * (2 * G.He + G.hop) * 4);
  for (int k = 0; k < G.nf; ++k) zero(S.zf[k], (int64_t)(2 * G.Hf + G.hop) * c.inter_channels * 4);
  zero(S.zd, (int64_t)(2 * G.Hd1 + G.hop) * c.inter_channels * 4);
  for (int j = 0; j < 3; ++j) zero(S.s0[j], (int64_t)(2 * G.Hd2 + G.hop) * G.r0 * P.stages[0].ch * 2);
  ok = ok && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pos_dev + slot), 0, 1, st) == hipSuccess;
  ok = ok && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(len_dev + slot), length, 1, st) == hipSuccess;   // no host buffer involved
  return ok ? QVC_OK : QVC_ERR_LAUNCH;
}

int qvc_infer_batch_timed(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* g,
                          const float* noise, float* out, int32_t batch, int32_t frames, void* workspace,
                          int64_t workspace_bytes, void* stream, qvc_launch_record* records, int32_t max_records,
                          int32_t* n_records) {
  if (!unit || !g || !noise || !out || !records || max_records <= 0 || !n_records) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  TimedBackend be{static_cast<hipStream_t>(stream), records, max_records};
  Path<TimedBackend> c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.cond_table(g);
  c.enc_p(unit, noise, c.wsp<float>(W.z));
  c.flow(c.wsp<float>(W.z));
  c.dec_trunk_wave(c.wsp<float>(W.z), c.wsp<float>(W.post), out);
  const int fin = be.finish();
  *n_records = be.n;
  return c.status != QVC_OK ? c.status : fin;
}

int qvc_enc_p(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* noise, float* z_p_fm,
              int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!unit || !noise || !z_p_fm) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.enc_p(unit, noise, z_p_fm);
  return c.status;
}

int qvc_wn_stack(const qvc_config* cfg, const void* blob_dev, int32_t which, const float* x_fm, const float* g,
                 float* out_fm, int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x_fm || !out_fm || which < 0 || (which > 0 && !g)) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  if (which > cfg->n_flows) return QVC_ERR_BAD_ARG;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  const size_t bytes = (size_t)batch * frames * cfg->hidden_channels * 4;
  if (hipMemcpyAsync(c.wsp<float>(W.xw), x_fm, bytes, hipMemcpyDeviceToDevice, be.stream) != hipSuccess) return QVC_ERR_LAUNCH;
  if (which == 0) {
    c.wn(P.enc_wn, reinterpret_cast<const float*>(static_cast<const char*>(blob_dev) + P.enc_wn.inbias_off), 0);
  } else {
    const FlowStepPlan* f = nullptr;
    for (const FlowStepPlan& s : P.flow) if (s.layer == which - 1) f = &s;
    if (!f) return QVC_ERR_BAD_ARG;
    c.cond_table(g);
    c.wn(f->wn, c.wsp<float>(W.bb) + f->cond_row0, P.cond_rows);
  }
  if (c.status != QVC_OK) return c.status;
  if (hipMemcpyAsync(out_fm, c.wsp<float>(W.oacc), bytes, hipMemcpyDeviceToDevice, be.stream) != hipSuccess) return QVC_ERR_LAUNCH;
  return QVC_OK;
}

int qvc_flow_reverse(const qvc_config* cfg, const void* blob_dev, float* z_fm, const float* g, int32_t batch,
                     int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!z_fm || !g) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.cond_table(g);
  c.flow(z_fm);
  return c.status;
}

int qvc_flow_forward(const qvc_config* cfg, const void* blob_dev, float* z_fm, const float* g, int32_t batch,
                     int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!z_fm || !g) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.cond_table(g);
  c.flow(z_fm, /*forward=*/true);
  return c.status;
}

int qvc_enc_q(const qvc_config* cfg, const void* encq_blob_dev, const float* spec, const float* g, const float* noise,
              float* z_fm, int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!spec || !g || !noise || !z_fm) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, encq_blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  const EncQPlan Q = build_encq_plan(*cfg);
  if (Q.status != QVC_OK) return Q.status;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(encq_blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.enc_q(Q, static_cast<const char*>(encq_blob_dev), spec, g, noise, z_fm);
  return c.status;
}

int qvc_dec_trunk(const qvc_config* cfg, const void* blob_dev, const float* z_fm, const float* g, float* post_fm,
                  int32_t batch, int32_t frames, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!z_fm || !g || !post_fm) return QVC_ERR_BAD_ARG;
  Plan P; Workspace W;
  int st = check_common(cfg, blob_dev, batch, frames, workspace, workspace_bytes, P, W);
  if (st != QVC_OK) return st;
  HipBackend be; be.stream = be.stream0 = static_cast<hipStream_t>(stream);
  Ctx c{P, static_cast<const char*>(blob_dev), static_cast<char*>(workspace), W, batch, frames, be};
  c.cond_table(g);
  c.dec_trunk(z_fm, post_fm);
  return c.status;
}

int qvc_istft_synth(const qvc_config* cfg, const void* blob_dev, const float* post_fm, float* out, float* y_mb,
                    int32_t batch, int32_t post_frames, void* stream) {
  if (!cfg || !blob_dev || !post_fm || !out || batch <= 0 || post_frames <= 1) return QVC_ERR_BAD_ARG;
  Plan P = build_plan(*cfg);
  if (P.status != QVC_OK) return P.status;
  TailArgs ta{post_fm, reinterpret_cast<const float*>(static_cast<const char*>(blob_dev) + P.fir_off), out, y_mb, batch,
              post_frames};
  return launch_tail(ta, stream);
}

int64_t qvc_conv1d_scratch_bytes(int32_t cout, int32_t cin, int32_t k) {
  if (cout <= 0 || cin <= 0 || k <= 0 || cout % 4 || cin % 8) return QVC_ERR_BAD_ARG;
  ConvDesc d = make_conv(cout, cin, k, 1);
  return align_up(d.w_bytes(), 256) + align_up(d.b_bytes(), 256);
}

int64_t qvc_conv1d_workspace_bytes(int32_t batch, int32_t cout, int32_t cin, int32_t frames) {
  if (batch <= 0 || cout <= 0 || cin <= 0 || frames <= 0) return QVC_ERR_BAD_ARG;
  return align_up((int64_t)batch * frames * cout * 4, 256);
}

int qvc_conv1d(const float* x, const float* w_host, const float* bias_host, float* y, int32_t batch, int32_t cin,
               int32_t cout, int32_t frames, int32_t k, int32_t dilation, float slope_in, int32_t operand_dtype,
               void* w_scratch_host, void* w_scratch_dev, int64_t scratch_bytes, void* workspace,
               int64_t workspace_bytes, void* stream) {
  if (!x || !w_host || !y || !w_scratch_host || !w_scratch_dev || !workspace) return QVC_ERR_BAD_ARG;
  if (batch <= 0 || frames <= 0 || k <= 0 || k % 2 == 0 || dilation <= 0 || cout % 4 || cin % 8) return QVC_ERR_BAD_ARG;
  if (operand_dtype != QVC_BF16 && operand_dtype != QVC_F16) return QVC_ERR_BAD_ARG;
  ConvDesc d = make_conv(cout, cin, k, dilation);
  d.w_off = 0; d.b_off = align_up(d.w_bytes(), 256);
  if (scratch_bytes < d.b_off + align_up(d.b_bytes(), 256)) return QVC_ERR_SMALL_BUFFER;
  if (workspace_bytes < qvc_conv1d_workspace_bytes(batch, cout, cin, frames)) return QVC_ERR_SMALL_BUFFER;
  pack_plain_conv(d, w_host, bias_host, operand_dtype, static_cast<char*>(w_scratch_host));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(w_scratch_dev, w_scratch_host, (size_t)(d.b_off + d.b_bytes()), hipMemcpyHostToDevice, s) != hipSuccess)
    return QVC_ERR_LAUNCH;
  ConvArgs a;
  a.w = w_scratch_dev;
  a.bias = reinterpret_cast<const float*>(static_cast<const char*>(w_scratch_dev) + d.b_off);
  a.x = x; a.x_kind = XK_F32_CM; a.x_bs = (int64_t)cin * frames; a.x_ts = frames; a.T_in = frames; a.slope_in = slope_in;
  a.Nq = frames; a.T_out = frames;
  a.y32 = static_cast<float*>(workspace); a.y32_bs = (int64_t)frames * cout; a.y32_ts = cout;
  int st = launch_conv(d, a, batch, EPI_STD, operand_dtype, stream);
  if (st != QVC_OK) return st;
  return launch_fm_to_cm(static_cast<const float*>(workspace), y, batch, frames, cout, stream);
}

}  // extern "C"
