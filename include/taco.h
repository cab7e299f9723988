/*
 * taco.h - C ABI of libtaco_hip.so, the MI355X (gfx950) implementation of the
 * cadia-lvl/tacotron synthesis path.
 *
 * The reference has no FFI: its de-facto boundary is the single
 *     wav = session.run(self.wav_output, feed_dict={inputs, input_lengths})
 * at /root/reference/synthesizer.py:39-43 plus the model attributes
 * mel_outputs / linear_outputs / alignments (model/tacotron.py:53-56).  Each entry point
 * below replaces one slice of that graph and cites the reference code it stands for.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / HIP types in signatures
 *    (taco_stream_t is a hipStream_t passed as void*; NULL = the default stream).
 *  - All tensor pointers are DEVICE pointers owned by the caller, contiguous row-major,
 *    fp32 / int32, 16-byte aligned.  Weights handed to taco_create are HOST pointers.
 *  - Every call returns 0 on success, <0 on error; taco_last_error() gives the message
 *    (thread-local).  No exception crosses the boundary.
 *  - A handle is bound to one device and is not thread-safe: one handle per GPU/stream.
 *  - After taco_create the library allocates nothing: all scratch comes from the
 *    caller's workspace of at least taco_workspace_bytes() bytes.  Calls are
 *    asynchronous on `stream`; the workspace must stay alive until they finish.
 */
#ifndef TACO_H_
#define TACO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct taco_ctx* taco_handle;
typedef void* taco_stream_t;

/* Shapes of the graph; field-for-field the model/audio/eval values of
 * /root/reference/hparams.py:13-21,25-51,65-67 (hop/win already derived as
 * tools/audio.py:162-163 does). The first field carries the struct's size (see below). */
typedef struct TacoHParams {
    uint32_t struct_size;         /* = sizeof(TacoHParams) of the header the caller was built against; taco_create
                                     refuses any other value (fields are only ever appended: a caller with an older,
                                     shorter struct is told so instead of having bytes read past its end) */
    int32_t num_chars;            /* len(text.characters.chars) = 95, data/batch.py:38 */
    int32_t embedded_depth;       /* 256 */
    int32_t prenet_depth1;        /* 256 */
    int32_t prenet_depth2;        /* 128 */
    int32_t encoder_K;            /* 16 */
    int32_t encoder_bank_filters; /* 128 */
    int32_t encoder_proj1;        /* 128 */
    int32_t encoder_proj2;        /* 128 */
    int32_t encoder_proj_width;   /* 3 */
    int32_t encoder_highways;     /* 4 */
    int32_t encoder_highway_depth;/* 128 */
    int32_t encoder_gru_cells;    /* 128 */
    int32_t attention_depth;      /* 256 */
    int32_t decoder_depth;        /* 256 */
    int32_t num_mels;             /* 80 */
    int32_t outputs_per_step;     /* 5 */
    int32_t decoder_K;            /* 8 (post-CBHG) */
    int32_t decoder_bank_filters; /* 128 */
    int32_t decoder_proj1;        /* 256 */
    int32_t decoder_proj2;        /* 80 */
    int32_t decoder_proj_width;   /* 3 */
    int32_t decoder_highways;     /* 4 */
    int32_t decoder_highway_depth;/* 128 */
    int32_t decoder_gru_cells;    /* 128 */
    int32_t num_freq;             /* 1025 */
    int32_t n_fft;                /* 2048 */
    int32_t hop_length;           /* 300 */
    int32_t win_length;           /* 1200 */
    float   min_level_db;         /* -100 */
    float   ref_level_db;         /* 20 */
    float   power;                /* 1.5 */
    int32_t max_iters;            /* 200 */
    int32_t griffin_lim_iters;    /* 60 */
    /* max-pool of the CBHG (hparams.py:31-32,41-42 -> model/modules.py:70-71): any window (the reference's 2 is
     * fused into proj_1's operand loader, others run a kernel of their own), stride 1 only - another stride
     * shortens the pooled sequence and the residual of modules.py:80 cannot be formed (the reference graph does
     * not build either): taco_create refuses it instead of silently computing a different graph. */
    int32_t encoder_pooling_width;  /* 2 */
    int32_t encoder_pooling_stride; /* 1 */
    int32_t decoder_pooling_width;  /* 2 */
    int32_t decoder_pooling_stride; /* 1 */
} TacoHParams;

/* One TF variable of the inference graph: name relative to `model/inference/`
 * (the scopes opened at synthesizer.py:23 and model/tacotron.py:31), host data, shape. */
typedef struct TacoWeightEntry {
    const char*  name;
    const float* data;
    int32_t      ndim;
    int64_t      dims[4];
} TacoWeightEntry;

const char* taco_last_error(void);

/* Replaces graph construction + `saver.restore` (synthesizer.py:19-34): copies, folds
 * (batch-norm -> scale/shift) and re-packs the weights on `device`. */
int taco_create(const TacoHParams* hp, const TacoWeightEntry* weights, int32_t n_weights,
                int32_t device, taco_handle* out);
void taco_destroy(taco_handle h);

/* Upper bound of the scratch any entry point needs for these sizes. */
size_t taco_workspace_bytes(taco_handle h, int32_t B, int32_t T_in, int32_t max_iters);

/* Encoder: embedding lookup (data/batch.py:36-40) -> prenet (model/encoder.py:26-37,
 * model/modules.py:4-28, dropout = identity at inference) -> CBHG K=16
 * (model/encoder.py:39-63, model/modules.py:30-98; BiGRU masked by `lens`).
 * ids [B,T_in] int32 right-padded with 0, lens [B] int32 -> enc_out [B,T_in,2*gru]. */
int taco_encode(taco_handle h, const int32_t* ids, const int32_t* lens, int32_t B, int32_t T_in,
                float* enc_out, void* ws, size_t ws_bytes, taco_stream_t stream);

/* Autoregressive decoder loop: model/decoder.py:18-42 (AttentionWrapper(DecoderPrenetWrapper(
 * GRUCell), BahdanauAttention), ConcatOutputAndAttentionWrapper, MultiRNNCell, r-frame
 * OutputProjectionWrapper, dynamic_decode), model/rnn_wrappers.py:24-26,51-53,
 * model/helpers.py:47-85 (TestingHelper: <GO> frame, feed last of r frames, all-zero stop)
 * and the alignment transpose of model/tacotron.py:51.
 *   mel   [B, max_iters*r, num_mels]   (rows past steps_run*r are zero)
 *   align [B, T_in, max_iters]         (columns past steps_run are zero)
 *   steps_run: device int32[1], number of decoder steps dynamic_decode would have run.
 *   dropout_keep: NULL (inference: tf.layers.dropout is the identity, modules.py:27), or
 *   device float [max_iters, B, prenet_depth1 + prenet_depth2] of 0/1 keep flags replaying a
 *   training-mode mask; kept activations are scaled by 1/(1-0.5). */
int taco_decode(taco_handle h, const float* enc_out, int32_t B, int32_t T_in, int32_t max_iters,
                const float* dropout_keep, float* mel, float* align, int32_t* steps_run,
                void* ws, size_t ws_bytes, taco_stream_t stream);

/* The same loop teacher-forced, as the reference runs it with TrainingHelper (model/helpers.py:6-43,
 * model/decoder.py:38 when is_training): step 0 is fed the <GO> frame, step t > 0 the target frame
 * targets[:, t*r - 1, :] ("every r-th target frame", helpers.py:15,42) instead of the previous output; all n_steps
 * steps run, there is no stop test (helpers.py:41), steps_run receives n_steps.
 *   targets [B, n_steps*r, num_mels] device fp32; mel, align, dropout_keep as in taco_decode (the reference's
 *   training graph also has the prenet dropout active: replay its mask through dropout_keep, or pass NULL).
 * Runs the per-stage kernels (the persistent kernel feeds its own outputs back). */
int taco_decode_teacher(taco_handle h, const float* enc_out, int32_t B, int32_t T_in, int32_t n_steps,
                        const float* targets, const float* dropout_keep, float* mel, float* align,
                        int32_t* steps_run, void* ws, size_t ws_bytes, taco_stream_t stream);

/* Post-processing net: post-CBHG K=8 without sequence lengths + dense to num_freq
 * (model/decoder.py:44-57).  mel [B,T,num_mels] -> linear [B,T,num_freq]. */
int taco_postnet(taco_handle h, const float* mel, int32_t B, int32_t T, float* linear,
                 void* ws, size_t ws_bytes, taco_stream_t stream);

/* tools/audio.py:57-64 spectrogram_tensorflow_inv: de-normalise, dB->amplitude, ^power,
 * then Griffin-Lim with zero initial phase (audio.py:183-205; uncentred STFT, periodic Hann
 * on analysis and synthesis, no window-sum normalisation).  The reference inverts only
 * linear_outputs[0] (synthesizer.py:26); here every row of the batch is inverted.
 * linear [B,T,num_freq] -> wav [B,(T-1)*hop+win].  No de-emphasis (host, synthesizer.py:44). */
int taco_griffin_lim(taco_handle h, const float* linear, int32_t B, int32_t T, int32_t iters,
                     float* wav, void* ws, size_t ws_bytes, taco_stream_t stream);

/* taco_griffin_lim for callers that cannot allocate and bring no workspace. No counterpart in the reference (its
 * vocoder is part of the TF graph, tools/audio.py:57-64); it exists for code under HIP graph capture, tight serving
 * loops and harnesses that forbid hipMalloc. Same kernels, same results, bit for bit, as taco_griffin_lim; the working
 * set (magnitudes + two frame buffers) lives in a static device array of 20 MiB inside the library, one copy per GPU
 * the library runs on, held until the process ends, its address resolved at taco_create. Limit: B*T <= 1024 frames
 * (one utterance of 200 decoder steps x 5 frames); larger inputs are refused (<0, message). The call allocates nothing
 * (no hipMalloc / hipFree / event or stream creation). All calls on one GPU share that GPU's array: calls that overlap
 * in time on one GPU (two streams, two handles) race; serialise them. Nothing in the array has to start at zero. */
int taco_griffin_lim_noalloc(taco_handle h, const float* linear, int32_t B, int32_t T, int32_t iters,
                             float* wav, taco_stream_t stream);

/* tools/audio.py:44-55 spectrogram_inv up to, not including, pre_emphasis_inv (host): the
 * numpy/librosa flavour of Griffin-Lim (audio.py:167-181) that train.py uses for its evaluation
 * audio: centred, reflect-padded librosa.stft (audio.py:119-134) and librosa.istft with
 * window-sum-square normalisation (audio.py:136-148), phases exp(i angle(.)) (unit phase at 0).
 * phase_u [B,T,num_freq] holds the uniform [0,1) draws of audio.py:175 (np.random.rand, here
 * frame-major) or is NULL for zero initial phase.  linear [B,T,num_freq] -> wav [B,hop*(T-1)].
 * Needs hop*(T-1) > n_fft/2.  Workspace: as taco_griffin_lim (taco_workspace_bytes). */
int taco_griffin_lim_librosa(taco_handle h, const float* linear, const float* phase_u, int32_t B,
                             int32_t T, int32_t iters, float* wav, void* ws, size_t ws_bytes,
                             taco_stream_t stream);

/* Analysis side, what data/data_load.py:114-146 computes per utterance with tools/audio.py:23-42,66-79:
 *   D = librosa.stft(pre_emphasis(wav), n_fft, hop, win, 'hann')   (centred, reflect-padded; audio.py:37,109,119-134)
 *   linear = normalize(20 log10(max(1e-5, |D|)) - ref_level_db)               [B, T, num_freq]   (audio.py:40-42)
 *   mel    = normalize(20 log10(max(1e-5, mel_basis @ |D|)) - ref_level_db)   [B, T, num_mels]   (audio.py:78-79)
 * with T = 1 + n_samples / hop frames, time-major (the reference transposes its [freq, T] result the same way before
 * saving, data_load.py:130-131). wav [B, n_samples] device fp32, n_samples > n_fft/2; linear or mel may be NULL.
 * mel needs a weight entry "audio/mel_basis" [num_mels, num_freq] at taco_create (librosa.filters.mel, audio.py:246-252;
 * built by the host: tacotron_amd/tools/audio.py:build_mel_basis). Workspace: taco_spectrogram_workspace_bytes. */
size_t taco_spectrogram_workspace_bytes(taco_handle h, int32_t B, int32_t n_samples);
int taco_spectrogram(taco_handle h, const float* wav, int32_t B, int32_t n_samples, float preemphasis,
                     float* linear, float* mel, void* ws, size_t ws_bytes, taco_stream_t stream);

/* The whole session.run of synthesizer.py:39-43: encode -> decode -> postnet
 * [-> Griffin-Lim when wav != NULL].  enc_out may be NULL (kept in the workspace). */
int taco_synthesize(taco_handle h, const int32_t* ids, const int32_t* lens, int32_t B, int32_t T_in,
                    int32_t max_iters, int32_t gl_iters, float* enc_out, float* mel, float* linear,
                    float* align, float* wav, int32_t* steps_run,
                    void* ws, size_t ws_bytes, taco_stream_t stream);

/* Measurement hooks (no reference counterpart; used by bench.py for the roofline leg).
 * taco_timing_begin(kind) makes every later launch of kernel family `kind` (see
 * taco_kernel_name) record a HIP event pair on its stream; taco_timing_end synchronises,
 * returns launches and summed milliseconds, and switches timing off. */
int taco_timing_begin(taco_handle h, int32_t kind);
int taco_timing_end(taco_handle h, int32_t* launches, double* total_ms);
const char* taco_kernel_name(int32_t kind);
int32_t taco_num_kernel_kinds(void);

/* Selects how taco_decode runs the loop: 0 = one kernel per stage, eager launches; 1 = the same kernels
 * captured once per (B, T_in, max_iters, buffers) into a hipGraph and replayed; 2 (default) = one persistent
 * kernel for the whole loop (default layer sizes, no dropout replay, B <= 128; otherwise mode 1 is used). */
int taco_set_decode_graph(taco_handle h, int32_t mode);

/* Scheduling knob of the persistent decoder (no reference counterpart; TF's executor has nothing of the kind): the
 * number of CUs that persistent decodes of THIS process may hold on `device` at the same time (each workgroup of a
 * persistent decode owns a CU: 32 per 16 utterances). A decode that would exceed it waits, on its stream, for older
 * ones to finish. 0 = the device's CU count (default: as many decodes side by side as fit the chip, best when the
 * other kernels are short, i.e. mel + linear only); callers that also run the 60-iteration Griffin-Lim of other
 * batches beside the decodes get more out of the chip with one decode at a time (budget = 32 * ceil(B / 16)): the
 * vocoder's full-chip launches then always find three quarters of the CUs (measured 3.04 -> 3.28 M frames/s to the
 * waveform at B = 32, six batches in flight). A single decode larger than the budget is still admitted, alone. */
int taco_set_persist_budget(int32_t device, int32_t workgroups);
/* The value last set for `device` (0 if never set). The knob is per process and device, not per handle: a caller that
 * changes it for a job reads it first and puts it back afterwards (tacotron_amd/pipeline.py does). */
int32_t taco_get_persist_budget(int32_t device);

/* Scheduling knob of taco_griffin_lim / taco_synthesize (no reference counterpart): run the utterances of a batch as
 * `groups` (1..4, at least 8 utterances each) independent chains on streams of the handle's own, forked from and joined
 * to the caller's stream, so that the tail of one group's full-chip launch is filled by another group's. Same kernels
 * per utterance, same bits. Worth it for a batch that has the GPU to itself (B = 32: 6.2 -> 5.7 ms per 60 iterations);
 * with several batches in flight the tails are filled by the other batches anyway and the extra streams cost
 * throughput: default 1. Ignored under stream capture and while launches are being timed. */
int taco_set_gl_split(taco_handle h, int32_t groups);

/* Scheduling knob of the CBHG's bidirectional GRU (model/modules.py:92-98; no reference counterpart: TF picks its own
 * kernels). 0 (default): one workgroup scans one direction of ONE utterance (0.64 us per time step, 2 B workgroups of a
 * CU each: the latency path, and the only one for cell counts other than 128). 1: one workgroup scans one direction of
 * SIXTEEN utterances with the recurrent products on fp32 MFMA (16x16x4; 3.5 us per time step, B / 8 workgroups): a
 * quarter of the CU time per batch. 2: FOUR utterances per workgroup on the 4x4x1 MFMA (1.9 us per step, B / 2
 * workgroups). Every mode computes each utterance independently of what it is batched with, in a fixed summation order
 * of its own: results are batch-invariant within a mode and differ between the modes in the last bits (all are tested
 * against the oracle). Measured with six batches in flight (B = 32): 9.28 / 8.54 / 8.85 M frames/s for modes 0 / 1 / 2 -
 * the longer chain of a batch costs more than the freed CUs give back, so 0 is what the repository's callers use. */
int taco_set_bigru_mode(taco_handle h, int32_t mode);

/* Synchronous check of the last persistent decode (reads a few device words; call after the stream has been
 * synchronised): <0 if one of its bounded inter-workgroup waits timed out, in which case its outputs are invalid.
 * Failure handling of the persistent mode (no reference counterpart; the reference's loop is a tf.while_loop inside
 * session.run, synthesizer.py:39-43): a launch whose wait runs out sets a sticky word of the handle, so persistent
 * launches of that handle that are already queued return at once (one time-out, not one per queued launch); the next
 * taco_decode / taco_synthesize that finds the previous launch aborted (from a pinned host mirror, no synchronisation)
 * returns <0 with this message - the PREVIOUS call's outputs were invalid - and the handle runs mode 1 from then on,
 * until taco_set_decode_graph(h, 2) re-arms it. A batch with more workgroups than the device has CUs runs mode 1. */
int taco_decode_status(taco_handle h);

/* Test hook (no reference counterpart): sets the handle's sticky abort word as a timed-out persistent decode would,
 * so the failure path above can be exercised without hanging a GPU. */
int taco_debug_poison_decoder(taco_handle h);

/* Measurement hook (no reference counterpart): the price of ONE dependency edge of the persistent decoder's step on
 * `device`, right now: `rounds` dependent all-to-all hand-offs among 32 workgroups (one per CU), each publishing 1 KB of
 * {value, tag} granules and gathering all 32 KB before it may publish again - the transport of the decoder kernel with
 * no arithmetic in it. bench.py multiplies it by the 11 edges of a step for the decoder's dependency-chain roofline.
 * Allocates and frees its own few KB; synchronous. */
int taco_debug_handoff_probe(int32_t device, int32_t rounds, double* us_per_handoff);

/* Test hook (no reference counterpart): one launch of the dense / SAME-conv1d implicit-GEMM
 * kernel, C[m,n] = (act(sum_{j,c} X[b,t+j-padl,c] W[j*Cin+c,n] + bias[n])) * scale[n] + shift[n]
 * + R[m,n], m = b*T + t.  All pointers are device pointers; W is [taps*Cin, ldw] with ldw a
 * multiple of 128 (zero padded); bias/scale/shift/R may be NULL.  act: 0 none, 1 relu,
 * 2 sigmoid, 3 tanh; pool != 0 reads max(X[t], X[t+1]) (model/modules.py:70-71). */
int taco_debug_conv_gemm(const float* X, int32_t ldx, int32_t M, int32_t T, int32_t Cin, int32_t taps,
                         int32_t padl, const float* W, int32_t ldw, int32_t N, const float* bias,
                         const float* scale, const float* shift, const float* R, int32_t ldr, float* C,
                         int32_t ldc, int32_t act, int32_t pool, taco_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TACO_H_ */
