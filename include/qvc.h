/*
 * qvc.h -- C ABI of the MI355X-native QuickVC inference hot path (libqvc_hip.so).
 *
 * This is the drop-in boundary for the path SynthesizerTrn.infer =
 *   enc_p (WN) -> reverse ResidualCouplingBlock -> multi-stream iSTFT generator
 * of tarepan/QuickVC-official.  The reference has no native code; its "FFI" for
 * this path is the Python surface of models.py.  Each entry point below names the
 * reference interface (file:line under the reference checkout) it replaces.  The
 * Python host (quickvc-official_amd/engine.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types.
 *   - every function returns an int status: 0 = QVC_OK, negative = error
 *     (qvc_status_string() names it).  Nothing throws, aborts or prints.
 *   - device pointers are raw HIP device addresses owned by the caller
 *     (tensor.data_ptr()); the library allocates nothing and keeps no global state
 *     (two exceptions, neither reachable from the environment: a per-device flag that
 *     remembers the one-time opt-in to > 64 KiB of LDS, and the developer switches of
 *     qvc_debug_set(), which start at their production values).
 *   - all device work is enqueued asynchronously on the caller's hipStream_t
 *     (passed as void*); no host sync, no allocation, no host read inside, so
 *     every call is hipGraph-capturable.  Thread-safe by statelessness.
 *     Calls on DIFFERENT streams may overlap as long as each has its own workspace, state
 *     and output buffers (the packed weights are read-only and may be shared): bench.py
 *     and convert.py keep two batches in flight that way.
 *   - every entry point that needs scratch has a *_workspace_bytes() query.
 *
 * Layouts
 *   External tensors keep the reference's (B, C, T) contiguous fp32 layout.
 *   Internal activations are "frame-major": [B][T][C] (channel innermost), so that
 *   the 8 consecutive channels one MFMA lane consumes are one 16-byte access.
 */
#ifndef QVC_H
#define QVC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QVC_ABI_VERSION 8

/* ---- status codes ------------------------------------------------------- */
enum {
  QVC_OK = 0,
  QVC_ERR_BAD_ARG = -1,        /* null pointer, non-positive size, bad enum          */
  QVC_ERR_BAD_CONFIG = -2,     /* unsupported hyper-parameters                        */
  QVC_ERR_MISSING_TENSOR = -3, /* a state-dict entry the path needs is absent         */
  QVC_ERR_BAD_SHAPE = -4,      /* a tensor has an unexpected shape                    */
  QVC_ERR_SMALL_BUFFER = -5,   /* blob / workspace smaller than *_bytes() says        */
  QVC_ERR_LAUNCH = -6,         /* hipLaunchKernel / hipGetLastError reported failure  */
  QVC_ERR_NO_DEVICE = -7       /* no HIP device / wrong architecture                  */
};

/* ---- decoder variants (models.py:588) ----------------------------------- */
enum {
  QVC_DEC_ISTFT = 0,       /* iSTFT_Generator            models.py:98-192  (subbands = 1, no band synthesis) */
  QVC_DEC_MULTIBAND = 1,   /* Multiband_iSTFT_Generator  models.py:195-301 (fixed PQMF, pqmf.py:106-117)     */
  QVC_DEC_MULTISTREAM = 2  /* Multistream_iSTFT_Generator models.py:304-415 (learned 63-tap FIR) -- shipped  */
};

/* ---- MFMA operand types -------------------------------------------------- */
enum {
  QVC_BF16 = 0,  /* bf16 operands, fp32 accumulate (BASELINE.json configs 2-5) */
  QVC_F16 = 1,   /* fp16 operands, fp32 accumulate (same MFMA rate, 3 more mantissa bits) */
  QVC_BF16X = 2  /* mixed: bf16 MFMA operands in the fused ResBlock pairs (80 % of the path's FLOPs) with their residual
                    stream kept in fp16 (same bytes; the identity path of 18 chained pairs keeps 11 mantissa bits), fp16
                    everywhere else.  All-bf16 misses the 40 dB waveform bar (34 dB measured); this mode meets it */
};

#define QVC_MAX_UPS 4
#define QVC_MAX_RESBLOCKS 4

/*
 * Hyper-parameters of SynthesizerTrn (models.py:551-591) that shape the path.
 * Filled by the host from the reference's JSON "model" section.
 */
typedef struct qvc_config {
  int32_t unit_channels;            /* 256, hard-coded at models.py:579                 */
  int32_t inter_channels;           /* latent z channels                                 */
  int32_t hidden_channels;          /* WN width                                          */
  int32_t gin_channels;             /* speaker embedding size                            */
  int32_t wn_kernel_size;           /* 5  (models.py:582-584)                            */
  int32_t enc_layers;               /* 16 (models.py:583)                                */
  int32_t flow_layers;              /* 4  WN layers per coupling layer (models.py:584)   */
  int32_t n_flows;                  /* 4                                                  */
  int32_t upsample_initial_channel; /* 512                                               */
  int32_t n_ups;                    /* len(upsample_rates)                               */
  int32_t upsample_rates[QVC_MAX_UPS];
  int32_t upsample_kernel_sizes[QVC_MAX_UPS];
  int32_t n_resblocks;              /* len(resblock_kernel_sizes)                        */
  int32_t resblock_kernel_sizes[QVC_MAX_RESBLOCKS];
  int32_t resblock_dilations[QVC_MAX_RESBLOCKS][3];
  int32_t n_fft;                    /* gen_istft_n_fft  (16)                             */
  int32_t hop;                      /* gen_istft_hop_size (4)                            */
  int32_t subbands;                 /* 4 (1 for QVC_DEC_ISTFT)                           */
  int32_t decoder;                  /* QVC_DEC_*                                         */
  int32_t fir_taps;                 /* 63: multistream_conv_post / PQMF taps+1           */
  int32_t operand_dtype;            /* QVC_BF16 / QVC_F16                                */
  int32_t n_mel_channels;           /* SpeakerEncoder input size (models.py:508); 0 = 80  */
  int32_t spec_channels;            /* enc_q input size = filter_length/2+1 (convert.py:35); 0 = 641 */
} qvc_config;

/* One named fp32 tensor of the reference checkpoint's state_dict (utils.py:183-193). */
typedef struct qvc_tensor {
  const char* name;    /* e.g. "enc_p.enc.in_layers.3.weight_v"  */
  const float* data;   /* host pointer, contiguous fp32           */
  int32_t ndim;
  int64_t shape[4];
} qvc_tensor;

/* ---- library info ---------------------------------------------------------- */
int qvc_abi_version(void);
const char* qvc_status_string(int status);
/* 0 when a gfx950 device is visible to this process, QVC_ERR_NO_DEVICE otherwise. */
int qvc_device_check(void);

/* Developer / test switches (launch-shape and kernel-selection variants that must give identical results; the GPU
 * tests flip them in-process).  The library never reads environment variables: a switch changes only through this
 * call.  Names: "post_tail", "post_tail_nf", "pair_wide_launch", "pair_cm4", "conv_cl", "wn_chunk", "pair_chain3",
 * "wn_kernel".  Unknown name: QVC_ERR_BAD_ARG.  No reference counterpart. */
int qvc_debug_set(const char* name, int32_t value);
int qvc_debug_get(const char* name, int32_t* value);
/* fp32 -> f16 conversions that SATURATED (|x| > 65504) since the last reset, summed over all kernels.  Counted only
 * by the debug build libqvc_hip_sat.so (-DQVC_SATCOUNT; synchronises the device): an activation range the f16
 * streams cannot carry becomes a number instead of passing silently.  The product library spends no instruction on
 * it and returns QVC_ERR_BAD_CONFIG with *count = -1. */
int qvc_debug_saturations(int64_t* count, int32_t reset);

/* ---- weights: replaces nn.Module.load_state_dict + per-forward weight_norm ----
 * Reference: utils.py:148-180 (load), modules.py:54,64,67,134-143 and
 * models.py:327,333,346,357 (weight_norm recomputed every forward).  Folds
 * w = g*v/||v|| once, folds the four channel Flips (modules.py:165-170) into the
 * coupling layers' weight layout, rewrites ConvTranspose1d as polyphase filters,
 * converts to the MFMA operand type and lays everything out as per-wave
 * fragment streams.  Pure host code.
 */
int64_t qvc_blob_bytes(const qvc_config* cfg);
int qvc_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                     void* blob_host, int64_t blob_bytes);

/* ---- whole path: replaces SynthesizerTrn.infer after the speaker encoder ------
 * Reference: models.py:638-642 (z_p = enc_p(unit); z = flow(z_p, g, reverse);
 * o = dec(z, g)), generalised to a batch of utterances with per-utterance g.
 *   unit  (B, unit_channels, T) fp32   HuBERT-soft units          (convert.py:79)
 *   g     (B, gin_channels)     fp32   speaker embeddings          (models.py:635)
 *   noise (B, inter_channels, T) fp32  the N(0,1) draw of models.py:94
 *   out   (B, T*prod(ups)*hop*subbands) fp32 waveform = (B,1,320*T)
 */
int64_t qvc_workspace_bytes(const qvc_config* cfg, int32_t batch, int32_t frames);
/* Layout decisions the plan takes for `cfg` (diagnostics; no reference counterpart): info[0] enc_p.proj rows paired
 * [mu | log sigma] (sampling in the conv epilogue), [1] its fragments per wave, [2..3] up-samplers 0 / 1 lane-packed,
 * [4] conv_post + iSTFT / band synthesis can run as one launch, [5..6] waves per workgroup of the stage 0 / 1 ResBlock
 * pairs, [7] every ResBlock pair runs fused, three chains per launch (then qvc_aux branches are never taken). */
int qvc_plan_info(const qvc_config* cfg, int32_t info[8]);
int qvc_infer_batch(const qvc_config* cfg, const void* blob_dev,
                    const float* unit, const float* g, const float* noise, float* out,
                    int32_t batch, int32_t frames,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the same for a RAGGED batch: utterances of different lengths in one launch sequence ------------------
 * The reference converts one utterance of any length per call (convert.py:58-86, models.py:625-642); a batched
 * drop-in therefore has to take utterances of different lengths.  unit / noise are padded to max_frames
 * ((B, C, max_frames), the padding's content is ignored), frames_dev is a DEVICE array of `batch` int32 lengths
 * (1 < frames_dev[b] <= max_frames; not validated on the host: the call is asynchronous; values are clamped
 * to [0, max_frames] on the device), and every conv / WaveNet layer / iSTFT frame sees utterance b end at
 * frames_dev[b] (zero padding at ITS end, as in the reference).  out is (B, 320*max_frames): the first
 * 320*frames_dev[b] samples of row b equal the waveform of utterance b converted alone, the rest are zeros.
 * Same workspace as qvc_infer_batch(cfg, batch, max_frames).
 */
int qvc_infer_batch_ragged(const qvc_config* cfg, const void* blob_dev,
                           const float* unit, const float* g, const float* noise, float* out,
                           int32_t batch, int32_t max_frames, const int32_t* frames_dev,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* The same with the units as they are ON DISK: dataset/encode.py:33-38 saves each utterance's HuBERT-soft units as a
 * (frames, 256) float32 .npy, data_utils_new_new.py:121-122 transposes after loading.  unit_fm is [B][max_frames][256]
 * (frame-major, row b padded to max_frames; padding content ignored), so a batch of files can be read straight into
 * the upload buffer (include/qvc_io.h) with no host-side transpose. */
int qvc_infer_batch_ragged_fm(const qvc_config* cfg, const void* blob_dev,
                              const float* unit_fm, const float* g, const float* noise, float* out,
                              int32_t batch, int32_t max_frames, const int32_t* frames_dev,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* ---- streaming: one hop of new unit frames per call, state in a caller-owned buffer (BASELINE configs[4]) -----
 * The reference converts whole utterances; every op of the path is a bounded, symmetric
 * convolution, so the path also runs as a pipeline of segments (enc_p | each coupling layer | conv_pre+stage 0 |
 * stage 1+conv_post+iSTFT), each fed from a ring that keeps the last 2*H frames of its input: a frame costs
 * (2*H + hop)/hop of its offline cost per segment (1.08x at hop 320) and the output lags the input by
 * qvc_stream_lag_frames() frames -- the look-ahead a non-causal network needs.  Sequence starts and ends are
 * exact: every kernel masks rows by the window's absolute position.  csrc/qvc_stream.h has the details.
 *   state      caller-owned device buffer of qvc_stream_state_bytes(), ZEROED before a stream's first step
 *   unit_new   (B, unit_channels, hop) fp32: unit frames [pos[b], pos[b] + hop) (anything past the sequence end)
 *   noise_new  (B, inter_channels, hop) fp32: the N(0,1) draw of models.py:94 for frames
 *              [pos[b] - qvc_stream_noise_lag_frames(), ... + hop)  (the frames enc_p completes in this step)
 *   out        (B, hop * samples_per_frame) fp32: frames [pos[b] - lag, pos[b] - lag + hop); zeros outside [0, len[b])
 *   pos_dev, len_dev  device int32 [B]: first new frame of this step / sequence length (a large number while unknown).
 *              The caller advances pos by hop between steps (device arrays: a captured graph replays unchanged).
 * To flush a stream of length L keep stepping (inputs ignored) until pos - lag >= L.
 */
int64_t qvc_stream_state_bytes(const qvc_config* cfg, int32_t batch, int32_t hop);
int64_t qvc_stream_workspace_bytes(const qvc_config* cfg, int32_t batch, int32_t hop);
int32_t qvc_stream_lag_frames(const qvc_config* cfg);
int32_t qvc_stream_noise_lag_frames(const qvc_config* cfg);
int qvc_stream_step(const qvc_config* cfg, const void* blob_dev, void* state, int64_t state_bytes,
                    const float* unit_new, const float* g, const float* noise_new, float* out,
                    int32_t batch, int32_t hop, const int32_t* pos_dev, const int32_t* len_dev,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* Admit a NEW stream into slot `slot` of a running batch (a server whose streams start and end at different times):
 * zeroes that slot's rows of every ring in `state` and sets pos_dev[slot] = 0, len_dev[slot] = length (a large number
 * while unknown) on `stream`.  The other slots are untouched, a captured step graph is replayed unchanged (it only
 * holds pointers), and the caller copies the new stream's g row itself.  Ending a stream = writing its length into
 * len_dev[slot]; the slot is free again once pos - lag >= length.  No reference counterpart. */
int qvc_stream_reset_slot(const qvc_config* cfg, void* state, int64_t state_bytes, int32_t batch, int32_t hop,
                          int32_t slot, int32_t length, int32_t* pos_dev, int32_t* len_dev, void* stream);

/* ---- optional fork/join resources: lets the three independent ResBlocks of an MRF stage
 * (models.py:378-384) run as parallel branches (two auxiliary non-blocking streams + events), so a
 * memory-bound k=3 launch overlaps a compute-bound k=11 launch.  Created/destroyed by the caller
 * outside the hot call; with aux == NULL everything runs on `stream` in the same order.  Works under
 * hipGraph capture of `stream` (the branches become parallel graph nodes).
 */
typedef struct qvc_aux qvc_aux;
int qvc_aux_create(qvc_aux** out);
int qvc_aux_destroy(qvc_aux* aux);
int qvc_infer_batch_ex(const qvc_config* cfg, const void* blob_dev,
                       const float* unit, const float* g, const float* noise, float* out,
                       int32_t batch, int32_t frames,
                       void* workspace, int64_t workspace_bytes, void* stream, qvc_aux* aux);

/* ---- the same call with per-launch timing (diagnostics for bench.py's roofline leg) ----
 * Runs the identical launch sequence but brackets every launch with HIP events recorded on
 * `stream`, synchronises the stream at the end and fills `records` (at most max_records;
 * *n_records receives the number of launches).  Creates/destroys events, so unlike
 * qvc_infer_batch it is NOT hipGraph-capturable and must not be used in the timed region.
 */
typedef struct qvc_launch_record {
  char name[48];        /* e.g. "conv<f16,MF4,NF2,WM4,std>", "rbpair<bf16x,MF2,NF10,WM4>", "post_tail<f16>", "cond_gemv" */
  float ms;             /* device time of this launch (HIP events on the stream)                */
  double flops;         /* algorithmic FLOPs (2*MAC) of the launch, 0 for byte movers           */
  double bytes;         /* algorithmic bytes: activations in + out + weights, each counted once */
} qvc_launch_record;

int qvc_infer_batch_timed(const qvc_config* cfg, const void* blob_dev,
                          const float* unit, const float* g, const float* noise, float* out,
                          int32_t batch, int32_t frames,
                          void* workspace, int64_t workspace_bytes, void* stream,
                          qvc_launch_record* records, int32_t max_records, int32_t* n_records);

/* ---- stages (same workspace; for stage-level parity tests and profiling) ------
 * Frame-major fp32 tensors: z_p / z are [B][T][inter_channels].
 */
/* CondNormalWN.forward (enc_p), models.py:75-95: unit, noise -> z_p. */
int qvc_enc_p(const qvc_config* cfg, const void* blob_dev, const float* unit, const float* noise,
              float* z_p_fm, int32_t batch, int32_t frames,
              void* workspace, int64_t workspace_bytes, void* stream);
/* WN.forward, modules.py:69-114: one of the path's WaveNet stacks on its own.
 *   which = 0: enc_p.enc (enc_layers layers, g = NULL: no conditioning, modules.py:98)
 *   which = 1 + i: flow.flows[2*i].enc (flow_layers layers, conditioned on g through cond_layer, modules.py:83-96)
 *   x_fm [B][T][hidden] fp32 (the stack's input, i.e. the output of the `pre` 1x1)  ->  out_fm [B][T][hidden] fp32
 *   (the sum of the skip paths = WN's return value).  x_fm and out_fm must not overlap. */
int qvc_wn_stack(const qvc_config* cfg, const void* blob_dev, int32_t which, const float* x_fm, const float* g,
                 float* out_fm, int32_t batch, int32_t frames,
                 void* workspace, int64_t workspace_bytes, void* stream);
/* ResidualCouplingBlock.forward(reverse=True), models.py:39-51: in place on z. */
int qvc_flow_reverse(const qvc_config* cfg, const void* blob_dev, float* z_fm, const float* g,
                     int32_t batch, int32_t frames,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* Generator trunk, models.py:372-390: z -> subband_conv_post output [B][F][subbands*2*(n_fft/2+1)],
 * F = frames*prod(ups)+1 (the ReflectionPad1d((1,0)) adds one frame). */
int qvc_dec_trunk(const qvc_config* cfg, const void* blob_dev, const float* z_fm, const float* g,
                  float* post_fm, int32_t batch, int32_t frames,
                  void* workspace, int64_t workspace_bytes, void* stream);
/* exp / pi*sin / per-band iSTFT / zero-stuff / synthesis FIR, models.py:394-406 and
 * pqmf.py:106-117: post_fm [B][F][...] -> out (B, subbands*hop*(F-1)); y_mb (optional,
 * may be NULL) receives the sub-band signals (B, subbands, hop*(F-1)). */
int qvc_istft_synth(const qvc_config* cfg, const void* blob_dev, const float* post_fm,
                    float* out, float* y_mb, int32_t batch, int32_t post_frames, void* stream);

/* ---- one conv through the MFMA kernel (unit tests of the kernel itself) ---------
 * y[b,co,t] = bias[co] + sum_{ci,j} w[co,ci,j] * lrelu(x[b,ci,t+(j-(k-1)/2)*dil], slope_in)
 * x, y are (B,C,T) fp32; w is (Cout,Cin,k) fp32 on the HOST (packed internally into
 * w_scratch_host, then copied to w_scratch_dev on the stream).  Replaces
 * torch.nn.functional.conv1d as used at modules.py:91,104,150-153.
 */
int64_t qvc_conv1d_scratch_bytes(int32_t cout, int32_t cin, int32_t k);
int64_t qvc_conv1d_workspace_bytes(int32_t batch, int32_t cout, int32_t cin, int32_t frames);
int qvc_conv1d(const float* x, const float* w_host, const float* bias_host, float* y,
               int32_t batch, int32_t cin, int32_t cout, int32_t frames,
               int32_t k, int32_t dilation, float slope_in, int32_t operand_dtype,
               void* w_scratch_host, void* w_scratch_dev, int64_t scratch_bytes,
               void* workspace, int64_t workspace_bytes, void* stream);

/* ---- speaker encoder: replaces SpeakerEncoder.embed_utterance (models.py:507-546) -------
 * The step of SynthesizerTrn.infer in front of the path (models.py:635):
 * 3-layer LSTM n_mel -> gin over 128-frame partials at hop 64 (plus the last 128 frames),
 * Linear + ReLU + L2 normalisation per partial, mean over the partials (not re-normalised).
 * All partials of all utterances run in one persistent-recurrence launch per layer.
 *   mel (U, n_mel, mel_frames) fp32 -- the layout convert.py:77 hands to infer()
 *   g   (U, gin_channels)     fp32
 * Own blob (state-dict keys enc_spk.lstm.{weight,bias}_{ih,hh}_l{0,1,2}, enc_spk.linear.*),
 * so a host may pack it only when it embeds speakers.  Needs gin_channels % 8 == 0 and
 * gin_channels <= 256 (QVC_ERR_BAD_CONFIG otherwise).
 */
int64_t qvc_spk_blob_bytes(const qvc_config* cfg);
int qvc_spk_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                         void* blob_host, int64_t blob_bytes);
int64_t qvc_spk_workspace_bytes(const qvc_config* cfg, int32_t utterances, int32_t mel_frames);
int qvc_speaker_embed(const qvc_config* cfg, const void* spk_blob_dev, const float* mel, float* g,
                      int32_t utterances, int32_t mel_frames,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- mel front-end: replaces mel_processing.wave_to_mel (mel_processing.py:15-98) ---------
 * The step in front of the speaker encoder on the target side (convert.py:75-77):
 * reflect pad (n_fft-hop)/2, Hann STFT (center=False, win = n_fft), sqrt(re^2+im^2+1e-6),
 * mel-basis matmul, log(clamp(., 1e-5)).  All fp32: the STFT is a GEMM against a windowed DFT table
 * on the f32 MFMA (bitwise an fmaf chain), so it needs no FFT plan and no complex temporaries.
 *   wave (U, samples) fp32 in [-1, 1]   ->   mel (U, n_mels, frames) fp32,
 *   frames = (samples + 2*((n_fft-hop)/2) - n_fft) / hop + 1.
 * The mel filter bank itself is the caller's (n_mels x (n_fft/2+1), librosa.filters.mel in the
 * reference): qvc_mel_pack_tables turns it and the windowed DFT matrix into the device table.
 * Needs n_fft % 16 == 0, hop % 4 == 0, n_fft >= hop, samples > (n_fft-hop)/2.
 */
int64_t qvc_mel_table_bytes(int32_t n_fft, int32_t n_mels);
int qvc_mel_pack_tables(int32_t n_fft, int32_t hop, int32_t n_mels, const float* mel_basis_host,
                        void* table_host, int64_t table_bytes);
int64_t qvc_mel_workspace_bytes(int32_t n_fft, int32_t hop, int32_t utterances, int32_t samples);
int qvc_wave_to_mel(const void* table_dev, int32_t n_fft, int32_t hop, int32_t n_mels,
                    const float* wave, float* mel, int32_t utterances, int32_t samples,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- posterior direction: enc_q and the forward flow (models.py:617-618) -------
 * z ~ enc_q(spec | g) = CondNormalWN with speaker conditioning (models.py:75-95,582), then
 * z_p = flow(z, g) = ResidualCouplingBlock.forward(reverse=False) (models.py:39-51; x1 <- m + x1,
 * modules.py:217).  Not used by convert.py; it is the analysis half of the model (reconstruction /
 * evaluation) and runs on the same WaveNet kernels.  enc_q has its own blob (state-dict keys enc_q.*);
 * the forward flow reuses the path's blob: every coupling layer sees the same channel flip in both
 * directions, so the flip-folded pre / post weights serve both.
 *   spec  (B, spec_channels, T) fp32   linear spectrogram        (mel_processing.py:15-58)
 *   g     (B, gin_channels)     fp32
 *   noise (B, inter_channels, T) fp32  the N(0,1) draw of models.py:94
 *   z_fm / z: frame-major [B][T][inter_channels] fp32 (as the stage entry points above)
 * Both use the workspace of qvc_workspace_bytes(cfg, batch, frames).
 */
int64_t qvc_encq_blob_bytes(const qvc_config* cfg);
int qvc_encq_pack_weights(const qvc_config* cfg, const qvc_tensor* tensors, int32_t n_tensors,
                          void* blob_host, int64_t blob_bytes);
int qvc_enc_q(const qvc_config* cfg, const void* encq_blob_dev, const float* spec, const float* g,
              const float* noise, float* z_fm, int32_t batch, int32_t frames,
              void* workspace, int64_t workspace_bytes, void* stream);
int qvc_flow_forward(const qvc_config* cfg, const void* blob_dev, float* z_fm, const float* g,
                     int32_t batch, int32_t frames,
                     void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QVC_H */
