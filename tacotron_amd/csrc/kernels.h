// Internal declarations shared by the HIP translation units of libtaco_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace taco {

// Kernel families, for taco_timing_begin / taco_kernel_name.
enum KernelKind : int {
    KK_CONV_GEMM = 0,   // conv_gemm_kernel<*>: dense / conv1d implicit GEMM on fp32 MFMA
    KK_HIGHWAY,         // (retired: the highway combine is conv_gemm's ACT_HIGHWAY_FUSED epilogue; never launched)
    KK_GATHER,          // embed_gather_kernel
    KK_BIGRU,           // bigru_scan_kernel (one launch scans all T steps)
    KK_DEC_GEMM,        // dec_gemm_kernel: decoder-step skinny GEMM
    KK_DEC_ATTN,        // dec_attention_kernel
    KK_DEC_MISC,        // decoder finalisation (transposes, stop test)
    KK_DEC_GRAPH,       // the whole captured decode loop (one hipGraphLaunch)
    KK_GL_MAG,          // spectrogram de-normalisation
    KK_GL_ITER,         // fused STFT -> phase -> iSTFT Griffin-Lim iteration
    KK_GL_OLA,          // final overlap-add
    KK_DEC_PERSIST,     // dec_persistent_kernel: the whole decode loop as one launch
    KK_DEC_TILE,        // dec_tile_kernel: the whole decode loop as one launch, one workgroup per 16-utterance tile
    KK_COUNT
};

// Optional event-pair recorder around launches of one kernel family.
struct Timing {
    int kind = -1;
    std::vector<hipEvent_t> ev;   // pooled; pairs [2i], [2i+1]
    size_t used = 0;
    void begin(int k, hipStream_t s);
    void end(int k, hipStream_t s);
};

// ACT_HIGHWAY_FUSED: the whole highway layer in the epilogue. W's columns come in blocks of 64 = [32 H | 32 T] of the
// same 32 units (so that one lane of the 32x32 MFMA layout holds H and T of a unit), R is the layer input x
// [M, N/2], C the layer output y = relu(H) * sigmoid(T) + x * (1 - sigmoid(T)) [M, N/2] (modules.py:100-123).
// ACT_DB_NORM: clip((20 log10(max(1e-5, v)) - p0 - p1) / -p1, 0, 1) with p0 = ref_level_db, p1 = min_level_db
// (tools/audio.py:78-79,217,264: amplitude -> dB -> normalised), for the mel projection of taco_spectrogram.
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_HIGHWAY_FUSED = 5, ACT_DB_NORM = 6 };

// C[m, n] = epi( sum_{j<taps} sum_{c<Cin} X[b, t+j-padl, c] * W[j*Cin+c, n] ), m = b*T + t.
// epi(v) = ((act(v + bias[n])) * scale[n] + shift[n]) + R[m, n]   (absent pieces skipped)
// One convolution of a grouped launch (grid.z): its own weights, taps and output column offset.
struct GemmGroup {
    const float* W; const float* bias; const float* scale; const float* shift;
    int taps, padl, coff, pad_;
};
struct GemmArgs {
    const float* X; int ldx;
    int M, T;
    int Cin, taps, padl;
    const float* W; int ldw;      // packed [taps*Cin, ldw], ldw multiple of 128, zero padded
    int N;
    const float* bias;
    const float* scale;
    const float* shift;
    const float* R; int ldr;
    float* C; int ldc;
    int act;
    int pool;                     // A operand is max(X[t], X[t+1]) (max_pooling1d 2/1 'same')
    const GemmGroup* groups;      // device table of ngroups convolutions over the same X (W/bias/scale/shift/taps/padl
    int ngroups;                  //   above are then ignored; all share Cin, N, ldw, act), or null
    float* partial;               // split-K scratch of splitk * M * roundup(N, 64) floats, or null
    int splitk;                   // > 1: K split over grid.z + a reduction/epilogue kernel (conv_gemm_pick_splitk)
    float p0, p1;                 // activation parameters (ACT_DB_NORM)
};
int conv_gemm_pick_splitk(int M, int N, int nk);

// The tail of a CBHG as ONE launch (model/modules.py:87-98 up to the scan): n_hw highway layers of depth 128 followed by
// the hoisted input projections of the BiGRU, xp = y @ Wr + br (N = 768). A workgroup keeps its rows' activations in LDS
// from layer to layer; only the weights stream. Same arithmetic per row as the per-layer conv_gemm launches it replaces.
struct CbhgTailArgs {
    const float* X; int ldx;          // [M, 128] highway input
    int M;
    const float* Whw[8]; const float* bhw[8]; int n_hw;   // [128, 256] in the [32 H | 32 T] column blocks, bias [256]
    const float* Wr; const float* br; int ldwr, Nr;       // [128, ldwr >= Nr], Nr = 768
    float* out; int ldo;              // [M, Nr]
};
void launch_cbhg_tail(const CbhgTailArgs& a, hipStream_t s, Timing* tm);
void launch_conv_gemm(const GemmArgs& a, hipStream_t s, Timing* tm);

// tf.layers.max_pooling1d(pool_size = width, strides = 1, padding = 'same') over time on [M = B*T, ld] rows
// (model/modules.py:70-71): out[t] = max(x[t - (width-1)/2 .. t + width-1 - (width-1)/2]) within the utterance.
void launch_maxpool_same(const float* x, float* out, int M, int T, int ld, int width, hipStream_t s, Timing* tm);

// out[m, :] = lut[clamp(ids[m]), :]  (width floats, multiple of 4)
void launch_embed_gather(const int32_t* ids, const float* lut, float* out, int M, int width, int num_rows,
                         hipStream_t s, Timing* tm);

// Bidirectional GRU(G) scan. xp [B*T, 6G] = x @ [fw gates|fw cand|bw gates|bw cand] + biases; lens may be null;
// out [B*T, 2G]. G = 128 (the reference's value): Pg/Pc are the register images of bigru_scan_kernel per direction;
// any other G: Pg [2][G][2G], Pc [2][G][G], the recurrent halves k-major (bigru_scan_generic_kernel).
constexpr int BIGRU_H = 128;
constexpr int BIGRU_THREADS = 512;
// Throughput variant for G = 128 (bigru_scan_mfma_kernel): sixteen utterances per workgroup on fp32 MFMA; Pm
// [2 directions][8 waves][96][64 lanes] is the A-operand register image (api.hip).
void launch_bigru_scan_mfma(const float* xp, const float* Pm, const int32_t* lens, float* out, int B, int T,
                            hipStream_t s, Timing* tm);
// ... and four utterances per workgroup on the 4x4x1 MFMA (bigru_scan_mfma4_kernel); Pm4 [2][8 waves][96][64 lanes].
void launch_bigru_scan_mfma4(const float* xp, const float* Pm4, const int32_t* lens, float* out, int B, int T,
                             hipStream_t s, Timing* tm);
void launch_bigru_scan(const float* xp, const float* Pg, const float* Pc, const int32_t* lens, float* out,
                       int B, int T, int G, hipStream_t s, Timing* tm);

// Decoder-step skinny GEMM on "Q layout" activations: a vector of dimension K for batch column b
// lives at act[(k/4)*Bpad + b] as float4 {k..k+3}; Bpad is a multiple of 16.
enum DecEpi : int { DE_LINEAR = 0, DE_RELU = 1, DE_GATES = 2, DE_CAND = 3, DE_OUT = 4 };
struct DecGemmArgs {
    const float* src0; int k16_0;     // first K segment (K0 = 16*k16_0)
    const float* src1; int k16_1;     // second K segment or null/0
    const float* Wp;                  // packed [N/16][K/16][64][4]
    int N;
    const float* bias;                // [N] or null
    int epi;
    int B, Bpad;
    float* out;                       // LINEAR/RELU/OUT: result; GATES: r*h; CAND: unused
    float* u;                         // GATES: writes u; CAND: reads u
    float* h;                         // GATES: reads h; CAND: reads + updates h in place
    const float* res; float* res_out; // CAND: res_out = h_new + res (optional)
    const float* keep; int keep_ld;   // RELU: optional dropout keep flags [B, keep_ld] (+col offset applied)
    int32_t* nz;                      // unused (the stop test scans the outputs afterwards)
};
void launch_dec_gemm(const DecGemmArgs& a, hipStream_t s, Timing* tm);

struct DecAttnArgs {
    const float* q;        // Q layout [A/4][Bpad][4]
    const float* keys;     // [B, T_in, A]
    const float* values;   // [B, T_in, Dv]
    const float* v;        // [A]
    float* ctx;            // Q layout [Dv/4][Bpad][4]
    float* align;          // [B, T_in] row for this step (contiguous per utterance)
    int B, Bpad, T_in, A, Dv;
};
void launch_dec_attention(const DecAttnArgs& a, hipStream_t s, Timing* tm);

// outQ [S][NR/4][Bpad][4] -> mel [B][S*NR] (zero past steps*NR); alignS [S][B][T_in] -> align [B][T_in][S].
void launch_dec_finalize(const float* outQ, int gran, const float* alignS, int32_t* nz, int S, int NR, int B, int Bpad,
                         int T_in, float* mel, float* align, int32_t* steps_run, hipStream_t s, Timing* tm,
                         bool stop_test = true);
// targets [B, S*r, M] -> teachQ [S][M/4][Bpad][4]: frame (st+1)*r - 1 of every utterance, the input of step st+1
void launch_dec_pack_teacher(const float* targets, float* teachQ, int S, int r, int M, int B, int Bpad, hipStream_t s);

// Tile-resident decoder (dec_tile.hip): one workgroup per 16-utterance tile runs all S steps; the stages are those of
// the per-stage path (api.hip decode_steps) as a table. Buffers are the Q-layout buffers of the per-stage path.
enum DecTileBuf : int { DTB_GO = 0, DTB_CTX, DTB_HATT, DTB_H1, DTB_H2, DTB_P1, DTB_P2, DTB_RH, DTB_U, DTB_Q, DTB_D0,
                        DTB_O1, DTB_O2, DTB_COUNT };
constexpr int DT_STAGES = 11;        // pre1, pre2, attG, attC, query, proj, g1G, g1C, g2G, g2C, out
constexpr int DT_QUERY_STAGE = 4;    // the attention follows this stage
struct DecTileStage {
    const float* Wp; const float* bias;   // packed [N/16][K/16][64][4]; bias [N] or null
    int src0, k16_0, src1, k16_1;         // K segments: buffer ids (src1 = -1: none). Stage 0's src0 is the previous frame
    int N, epi;                           // DecEpi
    int out;                              // LINEAR / RELU: result; GATES: r*h; OUT: ignored (the step's slice of outQ)
    int h;                                // GATES: reads h; CAND: reads + updates h in place
    int res, res_out;                     // CAND: res_out = h_new + res, or -1
};
struct DecTileArgs {
    DecTileStage st[DT_STAGES];
    float* buf[DTB_COUNT];
    float* region; unsigned region_bytes;   // the one allocation that holds every buffer above and outQ (K operands are
                                            //   read through a buffer resource over it)
    float* outQ;              // [S][NR/4][Bpad][4]
    const float* keys;        // [B, T_in, 256]
    const float* values;      // [B, T_in, 256]
    const float* attn_v;      // [256]
    float* alignS;            // [S][B][T_in]
    int B, Bpad, T_in, S, NR, M;
    int xcd0;                 // set by the launcher: XCD slot of tile 0 in this launch (placement only)
};
int launch_dec_tile(const DecTileArgs& a, hipStream_t s, Timing* tm);

// Persistent decoder (dec_persistent.hip), default layer sizes only. Offsets o_* are in floats from `region`,
// the contiguous, zero-filled block of granule buffers the workgroups exchange through (sc1 stores / loads):
// a vector of dimension K takes K*Bpad*2 floats; [h_att ; ctx] (512 wide), h1 and h2 are double-buffered by step parity,
// rh and u hold one 256-wide buffer per GRU (3 each), e holds B*T_in {score, tag} pairs, outQ S*400 wide.
struct DecPersistArgs {
    const float* Wp[11];      // packed weights per stage: pre1 (folded, K = 512), pre2, attG, attC, query, proj,
                              //   g1G, g1C (folded, K = 768), g2G, g2C, out
    const float* bias[11];    // or null
    const float* bias_pre1_go;  // prenet-1 bias of step 0 (<GO> frame: no folded output-projection bias)
    const float* attn_v;
    const float* keys;        // [B, T_in, 256]
    const float* values;      // [B, T_in, 256] (encoder outputs)
    float* region;
    unsigned region_bytes;
    unsigned o_hc, o_h1, o_h2, o_p1, o_p2, o_rh, o_u, o_q, o_d0, o_o1, o_o2, o_e, o_outQ;   // o_hc: 2 x [h_att ; ctx]
    float* alignS;            // [S][B][T_in]
    unsigned long long* sync; // handle-owned status words (PD_SYNC_* below); the first 128 bytes are zeroed per call
    int B, Bpad, T_in, S;
    int lds_keys, lds_vals;   // filled by the launcher: key rows per attention part staged in LDS, values staged or not
    unsigned long long* dbg;  // diagnostic builds (-DPD_STAMPS): s_memrealtime stamps of workgroup 0, step 5
    int xcd_pack;             // placement experiment (launcher, TACO_PD_XCD_PACK): see dec_persistent_kernel
};
int launch_dec_persistent(DecPersistArgs a, hipStream_t s, Timing* tm);
// measurement only: `rounds` dependent all-to-all granule hand-offs among 32 workgroups (dec_persistent.hip);
// region: 2 x 32 x 256 floats, zero-filled; abort_word: 64 zeroed unsigned; sink: 32 floats
int launch_handoff_probe(float* region, unsigned region_bytes, int rounds, unsigned* abort_word, float* sink,
                         hipStream_t s);
// Status words of a handle's persistent decodes, as unsigned indices into DecPersistArgs::sync. [0, 32) are per launch
// (zeroed by the launcher): abort flag (1 = a bounded wait ran out, 2 = the launch left at once because the handle's
// sticky word was set), then the workgroup and the tag of the first such wait, and the count of workgroups that
// started. PD_SYNC_STICKY lies beyond the per-launch block: set with the abort flag, cleared only by taco_set_decode_graph.
enum : int { PD_SYNC_ABORT = 1, PD_SYNC_WG = 2, PD_SYNC_TAG = 3, PD_SYNC_ARRIVED = 4, PD_SYNC_STICKY = 32 };

// Griffin-Lim
struct GLPlan {
    int n_fft, hop, win, nbins;       // 2048, 300, 1200, 1025
    const float* twiddle;             // [n_fft] complex (cos, -sin)(2*pi*j/n_fft), float2
    const float* window;              // [win] periodic Hann
    const float* lds_image;           // [win floats | FFT_TW + 512 complex]: window, pass-2/3 twiddles and split twiddles
                                      //   exactly as gl_iter2 keeps them in LDS (built once at taco_create), or null
    // Experiment switches (TACO_GL_ONE_FRAME / TACO_GL_FRAME_STATE / TACO_GL_RUN), read from the environment ONCE per
    // chain by gl_read_switches and handed down: a change of the environment between two launches of one chain cannot
    // hand one kernel's state to another.
    bool one_frame = false;           // keep the one-frame-per-wave kernel
    int frame_state = -1;             // -1 default rule, 1 frames always, 0 run records whenever the geometry allows
    int run = 0;                      // 0 default rule, 8 or 16 frames per workgroup
    bool prepared = false;            // the device's kernel attributes are known to be set (taco_create did): no set-up call
};
void gl_read_switches(GLPlan& p);
void launch_gl_mag(const float* lin, float* mag, size_t n, float min_db, float ref_db, float power,
                   hipStream_t s, Timing* tm);
// frames_in/out [B*T][win]; first != 0: inverse of (mag, zero phase) only. librosa: the numpy/librosa flavour of the
// loop (normalised, trimmed, reflect-padded signal between iterations; first synthesis from the phases
// exp(2 pi i phase_u), phase_u [B*T][nbins] or null = zero phase).
// state: null = the loop's state is the frames, in and out. Otherwise *state says what frames_in holds (0 = frames,
// 8 / 16 = run records of that many frames, the overlap-added signal as gl_iter2_kernel lays it out, which fits the
// same buffer) and receives what frames_out holds after the launch; hand it from launch to launch and to launch_gl_ola.
// All launches of a chain must be given the same B and T.
// Returns 0, or -1 (nothing launched) when *state names run records but the launch would read frames.
int launch_gl_iter(const GLPlan& p, const float* frames_in, const float* mag, float* frames_out, int B, int T,
                   int first, hipStream_t s, Timing* tm, bool librosa = false, const float* phase_u = nullptr,
                   int* state = nullptr);
std::vector<float> gl_lds_image(const std::vector<float>& twiddle, const std::vector<float>& window);
void gl_prepare_device();   // per-device kernel attributes of gl_iter (idempotent)
void launch_gl_ola(const GLPlan& p, const float* frames, float* wav, int B, int T, hipStream_t s, Timing* tm,
                   int state = 0);
// librosa.istft's output: wav [B][hop (T-1)], window-sum-square normalised, n_fft/2 trimmed from both ends
void launch_gl_ola_norm(const GLPlan& p, const float* frames, float* wav, int B, int T, hipStream_t s, Timing* tm);
// Centred STFT analysis (librosa.stft of the pre-emphasised signal, tools/audio.py:37,119-134): wav [B, n] ->
// frames T = 1 + n / hop; mag [B*T, ldm] = |D| (columns >= nbins untouched), lin [B*T, nbins] normalised dB or null.
void launch_stft_center(const GLPlan& p, const float* wav, int B, int n, int T, float preemph, float ref_db,
                        float min_db, float* mag, int ldm, float* lin, hipStream_t s, Timing* tm);

}  // namespace taco
