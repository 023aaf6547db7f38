// libvtamiq_hip.so: C ABI (include/vtamiq_hip.h) around the gfx950 kernels.
//
// One engine = one VTAMIQ model instance on one GPU: packed weights (16-bit hi[/lo] planes for the ViT GEMMs, fp32 for
// everything else), a workspace sized for the largest (B, N) seen, and the launch sequence of VTAMIQ.forward
// (modules/vtamiq/vtamiq.py:94-119) with both images of a pair batched as 2B sequences through ONE encoder pass
// (the reference runs two serial passes with the same weights, vtamiq.py:100-101).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vtamiq_hip.h"
#include "../../include/vtamiq_hip_fp8.h"
#include "kernels.h"

using namespace vtq;

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// fp8 mode: per-tensor power-of-two activation scales (value * scale is rounded to e4m3, |.| clamped to 448), one per
// quantisation point: the packed patches, and per layer the LayerNorm-1 output, the attention context, the LayerNorm-2 output and
// the GELU output.  They start at the round-2 constants below (oracle/fp8_oracle.py STATIC carries the same) and are replaced by
// CALIBRATED ones: the first forward of an engine (or vtq_fp8_calibrate on a batch of the caller's choice) measures max |value| at
// every point on that batch -- each producing kernel reports it (kernels.h Fp8Obs) and is run again once its scale is chosen, so
// everything downstream already sees the final operands -- and takes the largest power of two that maps it to <= 224 (a factor two
// of headroom for other batches; e4m3 is a floating format, so headroom costs no precision until values underflow 2^-9).
// vtq_config.options & VTQ_OPT_FP8_STATIC_SCALES keeps the constants.  A value that still exceeds 448 after scaling is clamped and raises bit 2 of the
// error word (vtq_input_errors).
constexpr float kSPatch = 256.0f, kSLn = 8.0f, kSAtt = 16.0f, kSGelu = 4.0f;
[[maybe_unused]] constexpr float kFp8Target = 224.0f;

// Softmax scale of MultiHeadSelfAttention (transformer.py:158-160: scores / sqrt(head_dim), head_dim = 64) times log2(e): with the
// 3-term attention the engine folds it into the query projection at weight ingestion, so that scores arrive in log2 units and the
// kernels' exponent is exp2(s - max) -- a subtraction that is exact for the row maximum at any magnitude (attention.hip prescale_q).
constexpr float kQLog2Scale = 0.125f * 1.4426950408889634f;

struct Slot {
    void* dst = nullptr;      // destination (fp32 copy) or bf16 hi plane (split)
    int64_t numel = 0;
    bool split = false;       // true: pack to 16-bit planes (engine->f16, engine->wpl), or to e4m3 rows + per-row scales (fp8 mode)
    int64_t plane = 0;        // elements between hi and lo plane
    float* scale = nullptr;   // fp8 mode: destination of the rows' inverse scales
    int64_t K = 0;            // row length of a split tensor
    int64_t Kp = 0;           // row pitch of its packed planes (> K: zero-padded to the GEMM's K tile; 0 = K)
    float mul = 1.0f;         // the tensor is multiplied by this on ingestion (query projection: kQLog2Scale)
    bool ignore = false;      // accepted and dropped: a parameter the forward never reads (adapter pairs other than pair 0)
    bool loaded = false;
};

struct Layer {
    void *wqkv, *wo, *w1, *w2;                 // 16-bit planes / e4m3 rows
    int64_t pqkv, po, p1, p2;                  // plane strides
    float *sqkv, *so, *s1, *s2;                // fp8 mode: per-output-channel inverse weight scales
    float *bqkv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b, *g1, *g2;
    // Adapter pair 0 (transformer.py:177-194, 260-269): site 0 after attention, site 1 after the MLP.  down: [Hq_pad, H] planes
    // (rows >= H/4 zero), up: [H, Hq_pad] planes (K zero-padded); Hq_pad = H/4 rounded up to the GEMM tile (256)
    void *ad_dn[2], *ad_up[2];
    int64_t pad_dn[2], pad_up[2];
    float *ad_bdn[2], *ad_bup[2];
};
// One linear stage of the DiffNet head as the skinny-MFMA kernel reads it: fp16 hi/lo planes [2][ceil16(N)][Kp] (Kp = K padded
// to the 32-deep k-step with zeros) + fp32 bias.  The head always runs the 3-term fp16 form, whatever the encoder's precision.
struct HeadLin { void* wp = nullptr; int64_t plane = 0; int N = 0, K = 0, Kp = 0; const float* b = nullptr; };
struct Rcab { float *slope, *w, *b, *wd, *bd, *wu, *bu, *wcat, *bcat; HeadLin cat, up; };   // wcat/bcat: [Wc ; Wd Wc], folded at load time
struct Rg { std::vector<Rcab> rcabs; float *w, *b; HeadLin tail; };

}  // namespace

struct vtq_engine {
    vtq_config cfg{};
    Num lin{0, 1}, att{0, 1};          // operand format of the linear layers / of attention (QK^T, PV)
    int f16 = 0, apl = 1, wpl = 1;     // element type of every plane; planes per activation / per weight tensor
    int dbg_stop = -1;                 // tests: leave the encoder after stage layer * 7 + k (vtq_debug_stop_after), -1 = never
    bool fp8 = false;                  // linear layers on e4m3 operands (MX-scaled MFMA, unit block scales): VTQ_PREC_FP8
    float s_patch = kSPatch;           //   activation scales (see kSPatch ...): patches, then per layer {LN1, attention, LN2, GELU}
    std::vector<float> s_ln1, s_att, s_ln2, s_gelu;
    bool fp8_static = false, fp8_calibrated = false, calibrating = false;
    bool fp8_installed = false;        //   the current scales came from vtq_fp8_set_scales: a weight reload keeps them
    float* amax_slot = nullptr;        //   device word the producers report max |value| into during a calibration forward
    float* spatch = nullptr;           //   inverse weight scales of the patch embedding
    int64_t PDp = 0;                   // patch_dim rounded up to the GEMM's K granule (row pitch of the packed patches / weight)
    int64_t Hqp = 0;                   // adapters: H / 4 rounded up to the GEMM tile (N of the down projection, K of the up projection)
    int H = 0, Mdim = 0, T = 0;
    std::vector<void*> allocs;
    std::unordered_map<std::string, Slot> slots;
    // ViT
    void* wpatch = nullptr; int64_t ppatch = 0;
    float *bpatch = nullptr, *cls = nullptr, *extra = nullptr, *pos_table = nullptr, *scale_table = nullptr, *encw = nullptr,
          *encb = nullptr;
    std::vector<Layer> layers;
    // head
    float* diff_gamma = nullptr;
    std::vector<Rg> rgs;
    float *qdw = nullptr, *qdb = nullptr, *p1w = nullptr, *p1b = nullptr, *p2a = nullptr, *p4w = nullptr, *p4b = nullptr;
    HeadLin qd, p1, p4;
    // workspace
    int capB = 0, capN = 0;
    int64_t rows_alloc = 0;
    float* x = nullptr;
    void *lnbuf = nullptr, *big = nullptr;
    int64_t ln_plane = 0, big_plane = 0;
    int *pidx = nullptr, *sidx = nullptr, *row_map = nullptr;
    float* hb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* hhid = nullptr;
    int r_alloc = 0;                     // rows of the skinny-stage plane buffers (multiple of 64)
    void *hp[2] = {nullptr, nullptr}, *ht = nullptr, *hq = nullptr;   // head planes (fp16 hi/lo): [r_alloc][H] x2, [r_alloc][hidp], [r_alloc][H/4]
    int64_t hp_plane = 0, ht_plane = 0, hq_plane = 0;
    int hidp = 0;
    void *tl = nullptr, *th = nullptr;   // CLS-tail planes (encoder format): [r_alloc][H], [r_alloc][M]
    int64_t tl_plane = 0, th_plane = 0;
    float *xcls = nullptr, *lncls = nullptr, *qcls = nullptr;   // CLS-only last layer (fp32 rows)
    bool cls_prune = true;
    bool fuse_ln = false;                // VTQ_OPT_FUSED_LAYERNORM: residual GEMMs carry the next LayerNorm in their epilogue (gemm_rowln.hip)
    int32_t* err_host = nullptr;         // pinned landing word of vtq_input_errors (a pageable destination goes through the runtime's staging path)
    int* err_flag = nullptr;             // device word (vtq_input_errors): bit 0 = a position outside [0, 1) was clamped, bit 1 = non-finite CLS difference
    std::vector<void*> ws_allocs;
    float* trace = nullptr;
    int iqa_token = 0;                          // vtamiq.py:57, 107-108: the token row the head consumes (0 = CLS, 1 .. = register tokens)
    // profiling
    uint32_t prof_mask = 0;
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> ev_used;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
};

namespace {

int dev_alloc(vtq_engine* e, void** p, size_t bytes, bool ws = false) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    (ws ? e->ws_allocs : e->allocs).push_back(*p);
    return 0;
}

int add_f32(vtq_engine* e, const std::string& name, float** p, int64_t numel) {
    if (dev_alloc(e, (void**)p, numel * sizeof(float))) return 1;
    Slot s; s.dst = *p; s.numel = numel;
    e->slots[name] = s;
    return 0;
}

// planes (or e4m3 rows) for a [rows_total, K] weight; sub-slot `name` covers rows [row0, row0 + rows)
int add_split(vtq_engine* e, const std::string& name, void* base, int64_t plane, int64_t row0, int64_t rows, int64_t K,
              float* scale_base = nullptr, int64_t Kp = 0) {
    if (Kp == 0) Kp = K;
    Slot s; s.dst = (char*)base + row0 * Kp * (e->fp8 ? 1 : 2); s.numel = rows * K; s.split = true; s.plane = plane; s.K = K; s.Kp = Kp;
    s.scale = scale_base ? scale_base + row0 : nullptr;
    e->slots[name] = s;
    return 0;
}

int alloc_planes(vtq_engine* e, void** p, int64_t* plane, int64_t numel) {
    *plane = numel;
    return dev_alloc(e, p, (size_t)numel * 2 * e->wpl);
}

int build(vtq_engine* e) {
    const vtq_config& c = e->cfg;
    const int64_t H = c.hidden_size, M = c.mlp_dim, PD = c.patch_dim;
    const std::string emb = "transformer.embeddings.";
    if (add_f32(e, emb + "cls_token", &e->cls, H)) return 1;
    if (c.num_extra_tokens > 0 && add_f32(e, emb + "extra_tokens", &e->extra, (int64_t)c.num_extra_tokens * H)) return 1;
    // the patch-embedding GEMM runs on K padded to 256 (two K tiles of every operand format): 768 as is, ViT-B/8's 192 -> 256
    e->PDp = round_up(PD, 256);
    e->Hqp = round_up(H / 4, 256);
    if (alloc_planes(e, &e->wpatch, &e->ppatch, H * e->PDp)) return 1;
    if (e->fp8 && dev_alloc(e, (void**)&e->spatch, H * sizeof(float))) return 1;
    add_split(e, emb + "patch_embeddings.weight", e->wpatch, e->ppatch, 0, H, PD, e->spatch, e->PDp);
    if (add_f32(e, emb + "patch_embeddings.bias", &e->bpatch, H)) return 1;
    if (add_f32(e, emb + "positional_embeddings.positional_embeddings", &e->pos_table, ((int64_t)c.pos_grid * c.pos_grid + 1) * H))
        return 1;
    if (c.num_scales > 1 && add_f32(e, emb + "scale_embeddings.scale_embeddings", &e->scale_table, ((int64_t)c.num_scales + 1) * H))
        return 1;
    const std::string enc = "transformer.encoder.";
    if (add_f32(e, enc + "encoder_norm.weight", &e->encw, H) || add_f32(e, enc + "encoder_norm.bias", &e->encb, H)) return 1;
    e->layers.resize(c.num_layers);
    for (int i = 0; i < c.num_layers; ++i) {
        Layer& L = e->layers[i];
        memset(&L, 0, sizeof L);
        const std::string p = enc + "layers." + std::to_string(i) + ".";
        if (alloc_planes(e, &L.wqkv, &L.pqkv, 3 * H * H) || alloc_planes(e, &L.wo, &L.po, H * H) ||
            alloc_planes(e, &L.w1, &L.p1, M * H) || alloc_planes(e, &L.w2, &L.p2, H * M))
            return 1;
        if (dev_alloc(e, (void**)&L.bqkv, 3 * H * sizeof(float))) return 1;
        if (e->fp8 && (dev_alloc(e, (void**)&L.sqkv, 3 * H * sizeof(float)) || dev_alloc(e, (void**)&L.so, H * sizeof(float)) ||
                       dev_alloc(e, (void**)&L.s1, M * sizeof(float)) || dev_alloc(e, (void**)&L.s2, H * sizeof(float))))
            return 1;
        const char* qkvn[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            add_split(e, p + "attn." + qkvn[j] + ".weight", L.wqkv, L.pqkv, j * H, H, H, L.sqkv);
            Slot s; s.dst = L.bqkv + j * H; s.numel = H;
            e->slots[p + "attn." + qkvn[j] + ".bias"] = s;
        }
        if (e->att.terms == 3) {       // 3-term attention takes Q in log2 units: (x W_q + b_q) * kQLog2Scale, folded into W_q and b_q
            e->slots[p + "attn.query.weight"].mul = kQLog2Scale;
            e->slots[p + "attn.query.bias"].mul = kQLog2Scale;
        }
        add_split(e, p + "attn.out.weight", L.wo, L.po, 0, H, H, L.so);
        add_split(e, p + "ffn.fc1.weight", L.w1, L.p1, 0, M, H, L.s1);
        add_split(e, p + "ffn.fc2.weight", L.w2, L.p2, 0, H, M, L.s2);
        if (add_f32(e, p + "attn.out.bias", &L.bo, H) || add_f32(e, p + "ffn.fc1.bias", &L.b1, M) ||
            add_f32(e, p + "ffn.fc2.bias", &L.b2, H) || add_f32(e, p + "attention_norm.weight", &L.ln1w, H) ||
            add_f32(e, p + "attention_norm.bias", &L.ln1b, H) || add_f32(e, p + "ffn_norm.weight", &L.ln2w, H) ||
            add_f32(e, p + "ffn_norm.bias", &L.ln2b, H))
            return 1;
        if (c.use_layer_scale && (add_f32(e, p + "ls1.gamma", &L.g1, H) || add_f32(e, p + "ls2.gamma", &L.g2, H))) return 1;
        if (c.num_adapters > 0) {
            const int64_t Hq = H / 4, Hqp = e->Hqp;
            for (int site = 0; site < 2; ++site) {
                const std::string q = p + "adapter" + std::to_string(site + 1) + ".adapter.";
                if (alloc_planes(e, &L.ad_dn[site], &L.pad_dn[site], Hqp * H) || alloc_planes(e, &L.ad_up[site], &L.pad_up[site], H * Hqp) ||
                    dev_alloc(e, (void**)&L.ad_bdn[site], Hqp * sizeof(float)))
                    return 1;
                HIP_TRY(hipMemset(L.ad_dn[site], 0, (size_t)Hqp * H * 2 * e->wpl));      // rows >= Hq: zero weights, zero bias -> gelu(0) = 0
                HIP_TRY(hipMemset(L.ad_bdn[site], 0, Hqp * sizeof(float)));
                add_split(e, q + "0.weight", L.ad_dn[site], L.pad_dn[site], 0, Hq, H);
                { Slot sb; sb.dst = L.ad_bdn[site]; sb.numel = Hq; e->slots[q + "0.bias"] = sb; }
                add_split(e, q + "2.weight", L.ad_up[site], L.pad_up[site], 0, H, Hq, nullptr, Hqp);
                if (add_f32(e, q + "2.bias", &L.ad_bup[site], H)) return 1;
            }
            for (int a = 3; a <= 2 * c.num_adapters; ++a) {          // pairs >= 1 exist in the state_dict; the forward never reads them
                const std::string q = p + "adapter" + std::to_string(a) + ".adapter.";
                const int64_t n[4] = {Hq * H, Hq, H * Hq, H};
                const char* nm[4] = {"0.weight", "0.bias", "2.weight", "2.bias"};
                for (int k = 0; k < 4; ++k) { Slot sg; sg.numel = n[k]; sg.ignore = true; e->slots[q + nm[k]] = sg; }
            }
        }
    }
    if (c.diff_scale && add_f32(e, "diff_scale.gamma", &e->diff_gamma, H)) return 1;
    if (c.calibrate) {
        const int64_t hid = c.ca_hidden;
        e->rgs.resize(c.num_rgs);
        for (int g = 0; g < c.num_rgs; ++g) {
            Rg& R = e->rgs[g];
            R.rcabs.resize(c.num_rcabs);
            for (int k = 0; k < c.num_rcabs; ++k) {
                Rcab& r = R.rcabs[k];
                const std::string p = "quality_decoder." + std::to_string(g) + ".body." + std::to_string(k) + ".body.";
                if (add_f32(e, p + "1.weight", &r.slope, 1) || add_f32(e, p + "2.weight", &r.w, H * H) ||
                    add_f32(e, p + "2.bias", &r.b, H) || add_f32(e, p + "4.conv_du.1.weight", &r.wd, hid * H) ||
                    add_f32(e, p + "4.conv_du.1.bias", &r.bd, hid) || add_f32(e, p + "4.conv_du.4.weight", &r.wu, H * hid) ||
                    add_f32(e, p + "4.conv_du.4.bias", &r.bu, H))
                    return 1;
                if (dev_alloc(e, (void**)&r.wcat, (size_t)(H + hid) * H * 4) || dev_alloc(e, (void**)&r.bcat, (size_t)(H + hid) * 4)) return 1;
            }
            const std::string p = "quality_decoder." + std::to_string(g) + ".body." + std::to_string(c.num_rcabs) + ".";
            if (add_f32(e, p + "weight", &R.w, H * H) || add_f32(e, p + "bias", &R.b, H)) return 1;
        }
        const std::string p = "quality_decoder." + std::to_string(c.num_rgs) + ".";
        if (add_f32(e, p + "weight", &e->qdw, H * H) || add_f32(e, p + "bias", &e->qdb, H)) return 1;
    }
    if (add_f32(e, "q_predictor.1.weight", &e->p1w, (H / 4) * H) || add_f32(e, "q_predictor.1.bias", &e->p1b, H / 4) ||
        add_f32(e, "q_predictor.2.weight", &e->p2a, 1) || add_f32(e, "q_predictor.4.weight", &e->p4w, H / 4) ||
        add_f32(e, "q_predictor.4.bias", &e->p4b, 1))
        return 1;
    // fp16 hi/lo planes of every head matrix (filled by pack_head after each weight load)
    auto head_lin = [&](HeadLin& L, int N, int K, const float* bias) {
        L.N = N; L.K = K; L.Kp = (int)round_up(K, 32); L.b = bias;
        L.plane = round_up(N, 16) * L.Kp;
        if (dev_alloc(e, &L.wp, (size_t)L.plane * 2 * 2)) return 1;
        return hipMemset(L.wp, 0, (size_t)L.plane * 2 * 2) == hipSuccess ? 0 : fail("hipMemset failed");
    };
    if (c.calibrate) {
        for (auto& R : e->rgs) {
            for (auto& r : R.rcabs)
                if (head_lin(r.cat, (int)H + c.ca_hidden, (int)H, r.bcat) || head_lin(r.up, (int)H, c.ca_hidden, r.bu)) return 1;
            if (head_lin(R.tail, (int)H, (int)H, R.b)) return 1;
        }
        if (head_lin(e->qd, (int)H, (int)H, e->qdb)) return 1;
    }
    if (head_lin(e->p1, (int)H / 4, (int)H, e->p1b) || head_lin(e->p4, 1, (int)H / 4, e->p4b)) return 1;
    e->hidp = (int)round_up(c.ca_hidden > 0 ? c.ca_hidden : 32, 32);
    return 0;
}

// fp32 matrices of the head -> the fp16 planes the skinny kernel streams (after the CA fold); enqueued on s
int pack_head(vtq_engine* e, hipStream_t s) {
    auto pack = [&](const HeadLin& L, const float* W) {
        HIP_TRY(launch_rows_to_planes(W, L.K, nullptr, L.wp, L.plane, L.Kp, L.N, L.K, 1, 2, s));
        return 0;
    };
    for (auto& R : e->rgs) {
        for (auto& r : R.rcabs) {
            HIP_TRY(launch_fold_ca(r.w, r.b, r.wd, r.bd, r.wcat, r.bcat, e->H, e->cfg.ca_hidden, s));
            if (pack(r.cat, r.wcat) || pack(r.up, r.wu)) return 1;
        }
        if (pack(R.tail, R.w)) return 1;
    }
    if (e->cfg.calibrate && pack(e->qd, e->qdw)) return 1;
    return pack(e->p1, e->p1w) || pack(e->p4, e->p4w);
}

struct Geometry {
    int S, S_pad, nseq;          // S_pad: row pitch of a sequence (= S: sequences are packed back to back)
    int64_t M_pad, P_pad, rows_alloc;
    SeqMap sm;
};

// nimg images per item: 2 = (ref, dist) FR pair, 3 = (ref, dist1, dist2) pairwise triplet
Geometry geometry(const vtq_engine* e, int B, int N, int nimg = 2) {
    Geometry g;
    g.S = N + e->T;
    // No per-sequence padding: attention masks keys >= S (its last 64-key tile and last 128-query block run into the next
    // sequence's rows, or into the tail / the 128 slack rows: finite values, never stored), every other kernel is
    // row-independent.  Only the whole batch is padded, to the GEMM tile height.
    g.S_pad = g.S;
    g.nseq = nimg * B;
    g.M_pad = round_up((int64_t)g.nseq * g.S_pad, 256);
    g.P_pad = round_up((int64_t)nimg * B * N, 256);
    g.rows_alloc = (g.M_pad > g.P_pad ? g.M_pad : g.P_pad) + 128;   // +128: attention over-read slack behind the last sequence
    g.sm = SeqMap{g.S_pad, g.nseq, (int)(g.M_pad - (int64_t)g.nseq * g.S_pad)};
    return g;
}

// capacity for up to `B` sequence pairs
int64_t capacity_rows(const vtq_engine* e, int B, int N) {
    const int64_t seq_rows = round_up((int64_t)2 * B * (N + e->T), 256);
    const int64_t patch_rows = round_up((int64_t)2 * B * N, 256);
    return (seq_rows > patch_rows ? seq_rows : patch_rows) + 128;
}

size_t workspace_bytes(const vtq_engine* e, int B, int N) {
    const int64_t rows = capacity_rows(e, B, N), P_pad = round_up((int64_t)2 * B * N, 256);
    const int64_t H = e->H, Wmax = (3 * H > e->Mdim ? 3 * H : e->Mdim);
    size_t b = 0;
    b += (size_t)rows * H * 4;                            // residual stream fp32
    b += (size_t)rows * H * 2 * e->apl;                   // LN / attention output planes
    b += (size_t)rows * Wmax * 2 * e->apl;                // qkv | mlp hidden | packed patches planes
    b += (size_t)P_pad * 4 * 3;                           // pos/scale indices, row map
    b += (size_t)2 * B * H * 4 * 6;                       // head ping-pong buffers (pairwise: 2 scores per item)
    b += (size_t)2 * B * 3 * H * 4;                       // CLS-only last-layer rows (fp32)
    const size_t ra = (size_t)round_up((int64_t)2 * B, 64);
    b += ra * (2 * H + e->hidp + H / 4) * 2 * 2;          // head planes (fp16 hi/lo)
    b += ra * (H + e->Mdim) * 2 * e->apl;                 // CLS-tail planes
    return b;
}

int reserve(vtq_engine* e, int B, int N) {
    if (B <= e->capB && N <= e->capN && e->x) return 0;
    const int nB = B > e->capB ? B : e->capB, nN = N > e->capN ? N : e->capN;
    HIP_TRY(hipDeviceSynchronize());
    for (void* p : e->ws_allocs) (void)hipFree(p);
    e->ws_allocs.clear();
    e->x = nullptr;
    struct { int64_t rows_alloc, P_pad; } g{capacity_rows(e, nB, nN), round_up((int64_t)2 * nB * nN, 256)};
    const int64_t H = e->H, Wmax = (3 * H > e->Mdim ? 3 * H : e->Mdim);
    e->rows_alloc = g.rows_alloc;
    e->ln_plane = g.rows_alloc * H;
    e->big_plane = g.rows_alloc * Wmax;
    if (dev_alloc(e, (void**)&e->x, (size_t)g.rows_alloc * H * 4, true) ||
        dev_alloc(e, &e->lnbuf, (size_t)e->ln_plane * 2 * e->apl, true) ||
        dev_alloc(e, &e->big, (size_t)e->big_plane * 2 * e->apl, true) ||
        dev_alloc(e, (void**)&e->pidx, (size_t)g.P_pad * 4, true) || dev_alloc(e, (void**)&e->sidx, (size_t)g.P_pad * 4, true) ||
        dev_alloc(e, (void**)&e->row_map, (size_t)g.P_pad * 4, true) ||
        dev_alloc(e, (void**)&e->hhid, (size_t)2 * nB * H * 4, true))
        return 1;
    for (int i = 0; i < 5; ++i)
        if (dev_alloc(e, (void**)&e->hb[i], (size_t)2 * nB * H * 4, true)) return 1;
    if (dev_alloc(e, (void**)&e->xcls, (size_t)2 * nB * H * 4, true) || dev_alloc(e, (void**)&e->lncls, (size_t)2 * nB * H * 4, true) ||
        dev_alloc(e, (void**)&e->qcls, (size_t)2 * nB * H * 4, true))
        return 1;
    {
        const int64_t ra = round_up((int64_t)2 * nB, 64);
        e->r_alloc = (int)ra;
        e->hp_plane = ra * H; e->ht_plane = ra * e->hidp; e->hq_plane = ra * (H / 4);
        e->tl_plane = ra * H; e->th_plane = ra * e->Mdim;
        if (dev_alloc(e, &e->hp[0], (size_t)e->hp_plane * 4, true) || dev_alloc(e, &e->hp[1], (size_t)e->hp_plane * 4, true) ||
            dev_alloc(e, &e->ht, (size_t)e->ht_plane * 4, true) || dev_alloc(e, &e->hq, (size_t)e->hq_plane * 4, true) ||
            dev_alloc(e, &e->tl, (size_t)e->tl_plane * 2 * e->apl, true) || dev_alloc(e, &e->th, (size_t)e->th_plane * 2 * e->apl, true))
            return 1;
        // rows >= R and the K-padding columns are read by the MFMA stages: zero once, never written
        HIP_TRY(hipMemset(e->hp[0], 0, (size_t)e->hp_plane * 4));
        HIP_TRY(hipMemset(e->hp[1], 0, (size_t)e->hp_plane * 4));
        HIP_TRY(hipMemset(e->ht, 0, (size_t)e->ht_plane * 4));
        HIP_TRY(hipMemset(e->hq, 0, (size_t)e->hq_plane * 4));
        HIP_TRY(hipMemset(e->tl, 0, (size_t)e->tl_plane * 2 * e->apl));
        HIP_TRY(hipMemset(e->th, 0, (size_t)e->th_plane * 2 * e->apl));
    }
    // finite contents everywhere: padded rows are computed on (never consumed) and must not breed NaNs
    HIP_TRY(hipMemset(e->x, 0, (size_t)g.rows_alloc * H * 4));
    HIP_TRY(hipMemset(e->lnbuf, 0, (size_t)e->ln_plane * 2 * e->apl));
    HIP_TRY(hipMemset(e->big, 0, (size_t)e->big_plane * 2 * e->apl));
    HIP_TRY(hipDeviceSynchronize());
    e->capB = nB;
    e->capN = nN;
    return 0;
}

struct Prof {
    vtq_engine* e; hipStream_t s; int cls; bool on; hipEvent_t a, b;
    Prof(vtq_engine* e_, hipStream_t s_, int cls_) : e(e_), s(s_), cls(cls_), on((e_->prof_mask >> cls_) & 1u) {
        if (!on) return;
        if (!e->ev_free.empty()) { a = e->ev_free.back().first; b = e->ev_free.back().second; e->ev_free.pop_back(); }
        else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(a, s);
    }
    ~Prof() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        e->ev_used.push_back({a, b, cls});
    }
};


// largest power of two s with m * s <= kFp8Target (exact: frexp, no log); 0 / non-finite m: keep `keep`
float fp8_pick_scale(float m, float keep) {
    if (!(m > 0.0f) || !std::isfinite(m)) return keep;
    int ex = 0;
    const float f = frexpf(m, &ex);                      // m = f * 2^ex, f in [0.5, 1);  224 = 0.875 * 2^8
    return ldexpf(1.0f, (f <= 0.875f ? 8 : 7) - ex);
}

// One fp8 producer stage (a kernel that writes e4m3 activation bytes with scale `sc`).  Normal forwards: run it once, saturation
// reported into the error word.  Calibration forward: run it reporting max |value|, read that (the one place a forward
// synchronises), choose `sc`, run it again with the final scale.  launch(scale, obs) must be idempotent.
template <typename F>
int fp8_stage(vtq_engine* e, hipStream_t s, float& sc, F launch) {
    if (!e->calibrating) return launch(sc, Fp8Obs{nullptr, e->err_flag});
    HIP_TRY(hipMemsetAsync(e->amax_slot, 0, 4, s));
    if (launch(sc, Fp8Obs{e->amax_slot, nullptr})) return 1;
    float m = 0.f;
    HIP_TRY(hipMemcpyAsync(&m, e->amax_slot, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    sc = fp8_pick_scale(m, sc);
    return launch(sc, Fp8Obs{nullptr, e->err_flag});
}

// All encoder layers for the g.nseq sequences, enqueued on s.
int run_encoder(vtq_engine* e, const Geometry& g, hipStream_t s, bool prune) {
    const vtq_config& c = e->cfg;
    const int H = e->H, Md = e->Mdim, T = e->T, L = c.num_layers, f16 = e->f16, apl = e->apl;
    const Num lin = e->lin;
    const int M = (int)g.M_pad;
    float* x = e->x;
    char* lnb = (char*)e->lnbuf;
    char* big = (char*)e->big;                               // QKV (ld 3H) and the MLP hidden (ld M) alias: never live together
    float *xcls = e->xcls, *lncls = e->lncls, *qcls = e->qcls;
    const int64_t trace_stride = (int64_t)g.nseq * T * H;
    // LayerNorm inside the residual GEMMs (gemm_rowln.hip): the out-proj launch also writes LayerNorm 2's planes, the fc2 launch the
    // NEXT layer's LayerNorm 1 planes; only layer 0's LayerNorm 1 (behind the patch embedding) is a launch of its own.  Not while a
    // test stops the encoder between stages (vtq_debug_stop_after addresses the separate stages).
    const bool fused = e->fuse_ln && e->dbg_stop < 0;
    auto resid_ln = [&](const void* A, int64_t a_plane, int lda, const void* W, int64_t w_plane, int K, const float* bias, const float* gamma,
                        const float* lnw, const float* lnbias) -> hipError_t {
        RowLnArgs a{};
        a.A = A; a.a_plane = a_plane; a.lda = lda; a.W = W; a.w_plane = w_plane; a.M = M; a.N = H; a.K = K; a.bias = bias; a.gamma = gamma; a.x = x;
        a.ln_w = lnw; a.ln_b = lnbias; a.out = lnb; a.o_plane = e->ln_plane;
        return launch_gemm_rowln(a, lin, s);
    };
    for (int i = 0; i < L; ++i) {
        const Layer& Ly = e->layers[i];
        if (prune && i == L - 1) {
            // ---- last layer: K/V for every row, everything else for the 2B CLS rows only (cls_tail.hip) ------------
            const int R = g.nseq;
            if (!(fused && i > 0)) { Prof p(e, s, VTQ_K_LN); HIP_TRY(launch_layernorm(x, Ly.ln1w, Ly.ln1b, lnb, e->ln_plane, M, H, f16, apl, s)); }
            {
                Prof p(e, s, VTQ_K_QKV);
                GemmArgs a{};
                a.A = lnb; a.a_plane = e->ln_plane; a.lda = H;
                a.W = (const char*)Ly.wqkv + (size_t)H * H * 2; a.w_plane = Ly.pqkv;
                a.M = M; a.N = 2 * H; a.K = H; a.bias = Ly.bqkv + H;
                a.out = big + (size_t)H * 2; a.o_plane = e->big_plane; a.ldo = 3 * H;
                HIP_TRY(launch_gemm(a, lin, EPI_BIAS, s));
            }
            {
                Prof p(e, s, VTQ_K_HEAD);
                // skinny MFMA stages on the R CLS rows (skinny.hip); activations between them as planes in the encoder's format
                auto stage = [&](const void* xa, int64_t xpl, int ldx, const void* W, int64_t wpl_, int N, int K, const float* bias) {
                    SkinnyArgs a{};
                    a.xa = xa; a.xa_plane = xpl; a.ldx = ldx; a.W = W; a.w_plane = wpl_; a.R = R; a.N = N; a.K = K; a.bias = bias;
                    a.ya_planes = apl;
                    return a;
                };
                const PlaneOut tl{e->tl, e->tl_plane, H, f16, apl, nullptr};       // every row kernel of the tail writes its consumer's planes
                HIP_TRY(launch_rows_ln(x + (int64_t)e->iqa_token * H, (int64_t)g.S_pad * H, Ly.ln1w, Ly.ln1b, lncls, xcls, R, H, tl, s));
                {   // query projection: rows 0 .. H-1 of the packed QKV weight
                    SkinnyArgs a = stage(e->tl, e->tl_plane, H, Ly.wqkv, Ly.pqkv, H, H, Ly.bqkv);
                    a.epi = SK_PLAIN; a.y = qcls; a.ldy = H; a.ycols = H;
                    HIP_TRY(launch_skinny(a, lin, s));
                }
                HIP_TRY(launch_cls_attention(qcls, big, e->big_plane, nullptr, R, g.S, g.S_pad, H, f16, apl, tl, s, e->att.terms == 3));
                {   // out-proj + LayerScale + residual, in place on the CLS rows
                    SkinnyArgs a = stage(e->tl, e->tl_plane, H, Ly.wo, Ly.po, H, H, Ly.bo);
                    a.epi = SK_RESID; a.gamma = Ly.g1; a.res = xcls; a.ldr = H; a.y = xcls; a.ldy = H; a.ycols = H;
                    HIP_TRY(launch_skinny(a, lin, s));
                }
                HIP_TRY(launch_rows_ln(xcls, H, Ly.ln2w, Ly.ln2b, lncls, nullptr, R, H, tl, s));
                {   // fc1 + GELU -> planes
                    SkinnyArgs a = stage(e->tl, e->tl_plane, H, Ly.w1, Ly.p1, Md, H, Ly.b1);
                    a.epi = SK_GELU; a.ya = e->th; a.ya_plane = e->th_plane; a.ldya = Md;
                    HIP_TRY(launch_skinny(a, lin, s));
                }
                {   // fc2 + LayerScale + residual
                    SkinnyArgs a = stage(e->th, e->th_plane, Md, Ly.w2, Ly.p2, H, Md, Ly.b2);
                    a.epi = SK_RESID; a.gamma = Ly.g2; a.res = xcls; a.ldr = H; a.y = xcls; a.ldy = H; a.ycols = H;
                    HIP_TRY(launch_skinny(a, lin, s));
                }
            }
            break;
        }
        // fp8 mode (VTQ_PREC_FP8): LayerNorm / attention / GELU outputs are e4m3 bytes with the static scales kS*, weights are
        // e4m3 rows with per-output-channel scales (de-scaled in the GEMM epilogue); the QKV output is one fp16 plane
        const bool f8m = e->fp8;
        // Adapter pair 0 (transformer.py:279-283): h <- h + up(gelu(down(h))) on the branch output BEFORE LayerScale and the residual
        // add.  The residual GEMM already added ls * h; the adapter's delta follows as two more GEMMs: down (N = H/4 padded to the
        // tile, GELU epilogue) into `tmp`, up (K = the padded H/4) with the same LayerScale into x.
        const bool adapters = c.num_adapters > 0;
        auto adapter_site = [&](const Layer& L, int site, void* hsrc, int64_t hplane, void* tmp, int64_t tplane, const float* gamma) -> hipError_t {
            GemmArgs d{};
            d.A = hsrc; d.a_plane = hplane; d.lda = H; d.W = L.ad_dn[site]; d.w_plane = L.pad_dn[site];
            d.M = M; d.N = (int)e->Hqp; d.K = H; d.bias = L.ad_bdn[site]; d.out = tmp; d.o_plane = tplane; d.ldo = (int)e->Hqp;
            hipError_t err = launch_gemm(d, lin, EPI_BIAS_GELU, s);
            if (err != hipSuccess) return err;
            GemmArgs u{};
            u.A = tmp; u.a_plane = tplane; u.lda = (int)e->Hqp; u.W = L.ad_up[site]; u.w_plane = L.pad_up[site];
            u.M = M; u.N = H; u.K = (int)e->Hqp; u.bias = L.ad_bup[site]; u.gamma = gamma; u.x = x;
            return launch_gemm(u, lin, EPI_RESID, s);
        };
        const int lnf = f8m ? 2 : f16, lnp = f8m ? 1 : apl;
        if (!(fused && i > 0)) {
            Prof p(e, s, VTQ_K_LN);
            if (f8m) {
                if (fp8_stage(e, s, e->s_ln1[i], [&](float sc, Fp8Obs ob) { HIP_TRY(launch_layernorm(x, Ly.ln1w, Ly.ln1b, lnb, e->ln_plane, M, H, 2, 1, s, sc, ob)); return 0; })) return 1;
            } else HIP_TRY(launch_layernorm(x, Ly.ln1w, Ly.ln1b, lnb, e->ln_plane, M, H, lnf, lnp, s));
        }
        if (e->dbg_stop == i * 7 + 0) return 0;
        {
            Prof p(e, s, VTQ_K_QKV);
            GemmArgs a{};
            a.A = lnb; a.a_plane = e->ln_plane; a.lda = H; a.W = Ly.wqkv; a.w_plane = Ly.pqkv;
            a.M = M; a.N = 3 * H; a.K = H; a.bias = Ly.bqkv; a.out = big; a.o_plane = e->big_plane; a.ldo = 3 * H;
            a.wscale = Ly.sqkv; a.ascale_inv = 1.0f / e->s_ln1[i];
            HIP_TRY(launch_gemm(a, lin, EPI_BIAS, s));
        }
        if (e->dbg_stop == i * 7 + 1) return 0;
        {
            Prof p(e, s, VTQ_K_ATTN);
            if (f8m) {
                if (fp8_stage(e, s, e->s_att[i], [&](float sc, Fp8Obs ob) { HIP_TRY(launch_attention(big, e->big_plane, lnb, e->ln_plane, g.nseq, g.S, g.S_pad, H, e->att, s, sc, ob, e->att.terms == 3)); return 0; })) return 1;
            } else HIP_TRY(launch_attention(big, e->big_plane, lnb, e->ln_plane, g.nseq, g.S, g.S_pad, H, e->att, s, 0.0f, Fp8Obs{nullptr, nullptr}, e->att.terms == 3));
        }
        if (e->dbg_stop == i * 7 + 2) return 0;
        {
            Prof p(e, s, VTQ_K_OUTPROJ);
            GemmArgs a{};
            a.A = lnb; a.a_plane = e->ln_plane; a.lda = H; a.W = Ly.wo; a.w_plane = Ly.po;
            a.M = M; a.N = H; a.K = H; a.bias = Ly.bo;
            a.wscale = Ly.so; a.ascale_inv = 1.0f / e->s_att[i];
            if (adapters) {                      // the branch output h itself, as operand planes for the adapter (QKV is consumed: `big` is free)
                GemmArgs hplanes = a;
                hplanes.out = big; hplanes.o_plane = e->big_plane; hplanes.ldo = H;
                HIP_TRY(launch_gemm(hplanes, lin, EPI_BIAS, s));
            }
            a.gamma = Ly.g1; a.x = x;
            if (fused) HIP_TRY(resid_ln(lnb, e->ln_plane, H, Ly.wo, Ly.po, H, Ly.bo, Ly.g1, Ly.ln2w, Ly.ln2b));   // x += ls1 * h; lnbuf = LayerNorm 2 (x)
            else
            HIP_TRY(launch_gemm(a, lin, EPI_RESID, s));              // x += ls1 * h
            if (adapters) HIP_TRY(adapter_site(Ly, 0, big, e->big_plane, lnb, e->ln_plane, Ly.g1));   // x += ls1 * (up(gelu(down(h))) )
        }
        if (e->dbg_stop == i * 7 + 3) return 0;
        if (!fused) {
            Prof p(e, s, VTQ_K_LN);
            if (f8m) {
                if (fp8_stage(e, s, e->s_ln2[i], [&](float sc, Fp8Obs ob) { HIP_TRY(launch_layernorm(x, Ly.ln2w, Ly.ln2b, lnb, e->ln_plane, M, H, 2, 1, s, sc, ob)); return 0; })) return 1;
            } else HIP_TRY(launch_layernorm(x, Ly.ln2w, Ly.ln2b, lnb, e->ln_plane, M, H, lnf, lnp, s));
        }
        if (e->dbg_stop == i * 7 + 4) return 0;
        {
            Prof p(e, s, VTQ_K_FC1);
            GemmArgs a{};
            a.A = lnb; a.a_plane = e->ln_plane; a.lda = H; a.W = Ly.w1; a.w_plane = Ly.p1;
            a.M = M; a.N = Md; a.K = H; a.bias = Ly.b1; a.out = big; a.o_plane = e->big_plane; a.ldo = Md;
            a.wscale = Ly.s1; a.ascale_inv = 1.0f / e->s_ln2[i];
            if (f8m) {
                if (fp8_stage(e, s, e->s_gelu[i], [&](float sc, Fp8Obs ob) { GemmArgs b = a; b.out_scale = sc; b.obs = ob; HIP_TRY(launch_gemm(b, lin, EPI_BIAS_GELU, s)); return 0; })) return 1;
            } else HIP_TRY(launch_gemm(a, lin, EPI_BIAS_GELU, s));
        }
        if (e->dbg_stop == i * 7 + 5) return 0;
        {
            Prof p(e, s, VTQ_K_FC2);
            GemmArgs a{};
            a.A = big; a.a_plane = e->big_plane; a.lda = Md; a.W = Ly.w2; a.w_plane = Ly.p2;
            a.M = M; a.N = H; a.K = Md; a.bias = Ly.b2;
            a.wscale = Ly.s2; a.ascale_inv = 1.0f / e->s_gelu[i];
            if (adapters) {                      // LayerNorm 2's planes are consumed: the branch output goes to `lnbuf`
                GemmArgs hplanes = a;
                hplanes.out = lnb; hplanes.o_plane = e->ln_plane; hplanes.ldo = H;
                HIP_TRY(launch_gemm(hplanes, lin, EPI_BIAS, s));
            }
            a.gamma = Ly.g2; a.x = x;
            if (fused) {                         // x += ls2 * h; lnbuf = the next layer's LayerNorm 1 (x) (none behind the last layer: final_diff normalises the CLS rows)
                const Layer* nx = (i + 1 < L) ? &e->layers[i + 1] : nullptr;
                HIP_TRY(resid_ln(big, e->big_plane, Md, Ly.w2, Ly.p2, Md, Ly.b2, Ly.g2, nx ? nx->ln1w : nullptr, nx ? nx->ln1b : nullptr));
            } else
            HIP_TRY(launch_gemm(a, lin, EPI_RESID, s));
            if (adapters) HIP_TRY(adapter_site(Ly, 1, lnb, e->ln_plane, big, e->big_plane, Ly.g2));
        }
        if (e->dbg_stop == i * 7 + 6) return 0;
        if (e->trace) HIP_TRY(launch_copy_tokens(x, e->trace + (i + 1) * trace_stride, g.nseq, g.sm, T, H, s));
    }
    return 0;
}

// PReLU slope of the head's first consumer (the first RCAB), or NULL when there is no decoder
const float* head_first_slope(const vtq_engine* e) { return (e->cfg.calibrate && !e->rgs.empty()) ? e->rgs[0].rcabs[0].slope : nullptr; }

// DiffNet head + quality predictor on d[HB][H] (the CLS difference after diff_scale; vtamiq.py:111-117,
// channel_attention.py:13-86) as a chain of skinny MFMA stages (skinny.hip), fp16 hi/lo operands, fp32 everywhere else.
//   RCAB (channel_attention.py:41-50, 77-86) = two stages with the CA squeeze folded into the conv (launch_fold_ca):
//     [c | t] = [Wc ; Wd Wc] prelu(r) + bcat, t = relu(.)        y = r + c * sigmoid(Wu t + bu)
//   Every stage writes the planes its consumer reads, already passed through the consumer's PReLU.
int run_head(vtq_engine* e, const float* d, int HB, float* q_out, hipStream_t s, bool planes_ready = false) {
    const vtq_config& c = e->cfg;
    const int H = e->H;
    const Num h3{1, 3};
    if (HB > e->r_alloc) return fail("run_head: %d rows exceed the reserved %d", HB, e->r_alloc);
    auto stage = [&](const void* xa, int64_t xpl, int ldx, const HeadLin& L) {
        SkinnyArgs a{};
        a.xa = xa; a.xa_plane = xpl; a.ldx = ldx; a.W = L.wp; a.w_plane = L.plane; a.R = HB; a.N = L.N; a.K = L.Kp; a.bias = L.b;
        a.ya_planes = 2;
        return a;
    };
    void *pin = e->hp[0], *pout = e->hp[1];
    if (!planes_ready) HIP_TRY(launch_rows_to_planes(d, H, head_first_slope(e), pin, e->hp_plane, H, HB, H, 1, 2, s));
    if (c.calibrate) {
        const float* xr = d;                 // residual-group input (fp32)
        float* xr_buf[2] = {e->hb[1], e->hb[0]};      // d lives in hb[0]: the first RG writes hb[1]
        float *y0 = e->hb[2], *y1 = e->hb[3], *cb = e->hb[4];
        const size_t ng = e->rgs.size();
        for (size_t gi = 0; gi < ng; ++gi) {
            Rg& R = e->rgs[gi];
            const float* y = xr;
            float* yo = y0;
            const size_t nr = R.rcabs.size();
            for (size_t k = 0; k < nr; ++k) {
                Rcab& r = R.rcabs[k];
                {
                    SkinnyArgs a = stage(pin, e->hp_plane, H, r.cat);
                    a.epi = SK_CONVCAT; a.nsplit = H; a.y = cb; a.ldy = H; a.ycols = H;
                    a.ya = e->ht; a.ya_plane = e->ht_plane; a.ldya = e->hidp; a.pcol0 = H;
                    HIP_TRY(launch_skinny(a, h3, s));
                }
                {
                    SkinnyArgs a = stage(e->ht, e->ht_plane, e->hidp, r.up);
                    a.epi = SK_GATE; a.res = y; a.aux = cb; a.ldr = H; a.y = yo; a.ldy = H; a.ycols = H;
                    a.ya = pout; a.ya_plane = e->hp_plane; a.ldya = H;
                    a.next_slope = (k + 1 < nr) ? R.rcabs[k + 1].slope : nullptr;      // the RG tail conv takes y itself
                    HIP_TRY(launch_skinny(a, h3, s));
                }
                y = yo;
                yo = (yo == y0) ? y1 : y0;
                std::swap(pin, pout);
            }
            {   // x + Conv1d(body(x))   (channel_attention.py:28-29; DropPath is identity in eval)
                float* xn = xr_buf[gi & 1];
                SkinnyArgs a = stage(pin, e->hp_plane, H, R.tail);
                a.epi = SK_RESID; a.res = xr; a.ldr = H; a.y = xn; a.ldy = H; a.ycols = H;
                a.ya = pout; a.ya_plane = e->hp_plane; a.ldya = H;
                a.next_slope = (gi + 1 < ng) ? e->rgs[gi + 1].rcabs[0].slope : nullptr;
                HIP_TRY(launch_skinny(a, h3, s));
                xr = xn;
                std::swap(pin, pout);
            }
        }
        {   // final Conv1d of the decoder (vtamiq.py:22): only the planes for the predictor are needed
            SkinnyArgs a = stage(pin, e->hp_plane, H, e->qd);
            a.epi = SK_PLAIN; a.ya = pout; a.ya_plane = e->hp_plane; a.ldya = H;
            HIP_TRY(launch_skinny(a, h3, s));
            std::swap(pin, pout);
        }
    }
    {   // q_predictor (vtamiq.py:71-77): Linear(H, H/4) -> PReLU -> Linear(H/4, 1)
        SkinnyArgs a = stage(pin, e->hp_plane, H, e->p1);
        a.epi = SK_PRELU; a.post_slope = e->p2a; a.ya = e->hq; a.ya_plane = e->hq_plane; a.ldya = H / 4;
        HIP_TRY(launch_skinny(a, h3, s));
        SkinnyArgs b = stage(e->hq, e->hq_plane, H / 4, e->p4);
        b.epi = SK_PLAIN; b.y = q_out; b.ldy = 1; b.ycols = 1;
        HIP_TRY(launch_skinny(b, h3, s));
    }
    return 0;
}

}  // namespace

extern "C" {

int vtq_abi_version(void) { return VTQ_ABI_VERSION; }
const char* vtq_last_error(void) { return g_err.c_str(); }

int vtq_create(const vtq_config* cfg, vtq_handle* out) {
    if (!cfg || !out) return fail("vtq_create: null argument");
    const vtq_config& c = *cfg;
    if (c.hidden_size != 768 && c.hidden_size != 1024) return fail("hidden_size %d unsupported (768 | 1024)", c.hidden_size);
    if (c.num_heads <= 0 || c.hidden_size / c.num_heads != 64 || c.hidden_size % c.num_heads)
        return fail("head_dim must be 64 (hidden %d, heads %d)", c.hidden_size, c.num_heads);
    if (c.mlp_dim % 256 || c.mlp_dim <= 0) return fail("mlp_dim %d must be a positive multiple of 256", c.mlp_dim);
    if (c.num_adapters < 0 || c.num_adapters > 64) return fail("num_adapters %d", c.num_adapters);
    if (c.num_adapters > 0 && c.precision == VTQ_PREC_FP8) return fail("adapters are not available in the fp8 mode (the adapter input would need an e4m3 copy)");
    if (c.patch_dim != 768 && c.patch_dim != 192) return fail("patch_dim %d unsupported (3*16*16 or 3*8*8)", c.patch_dim);
    if (c.num_layers < 1 || c.pos_grid < 1 || c.num_extra_tokens < 0) return fail("bad topology");
    if (c.options & ~(VTQ_OPT_FULL_LAST_LAYER | VTQ_OPT_FP8_STATIC_SCALES | VTQ_OPT_FUSED_LAYERNORM)) return fail("unknown vtq_config.options bits 0x%x", c.options);
    if ((c.options & VTQ_OPT_FUSED_LAYERNORM) && (c.mlp_dim % 32 || c.mlp_dim < 128)) return fail("VTQ_OPT_FUSED_LAYERNORM needs mlp_dim %% 32 == 0 and >= 128");
    if (c.calibrate && (c.num_rgs < 1 || c.num_rcabs < 1 || c.ca_hidden < 4 || c.ca_hidden % 4 || c.ca_hidden > 256))
        return fail("bad DiffNet topology (rgs %d, rcabs %d, ca_hidden %d)", c.num_rgs, c.num_rcabs, c.ca_hidden);
    vtq_engine* e = new vtq_engine();
    e->cfg = c;
    switch (c.precision) {
        case VTQ_PREC_BF16:   e->lin = Num{0, 1}; e->att = Num{0, 1}; break;
        case VTQ_PREC_BF16X3: e->lin = Num{0, 3}; e->att = Num{0, 3}; break;
        case VTQ_PREC_FP16:   e->lin = Num{1, 1}; e->att = Num{1, 1}; break;
        case VTQ_PREC_FP16X3: e->lin = Num{1, 3}; e->att = Num{1, 3}; break;
        case VTQ_PREC_FP16X2: e->lin = Num{1, 2}; e->att = Num{1, 3}; break;     // attention keeps the 3-term form
#ifdef VTQ_WITH_FP8
        case VTQ_PREC_FP8:    e->lin = Num{2, 1}; e->att = Num{1, 1}; e->fp8 = true; break;   // e4m3 linears, single-fp16 attention (2e-4 << the e4m3 step)
#else
        case VTQ_PREC_FP8:    delete e; return fail("precision 5 (fp8) is an experiment that this library was built without: build it with python -m vtamiq_amd.build --fp8 "
                                                    "(include/vtamiq_hip_fp8.h, vtamiq_amd/experimental_fp8.py)");
#endif
        default: delete e; return fail("unknown precision %d", c.precision);
    }
    e->f16 = e->fp8 ? 1 : e->lin.f16;          // fp8 mode: the QKV output is one fp16 plane (buffers are sized for it);
    e->apl = e->fp8 ? 1 : e->lin.apl();        //   LayerNorm / GELU / attention outputs are e4m3 bytes
    e->wpl = e->lin.wpl();
    e->H = c.hidden_size;
    e->Mdim = c.mlp_dim;
    e->T = 1 + c.num_extra_tokens;
    e->cls_prune = !(c.options & VTQ_OPT_FULL_LAST_LAYER);
    e->fuse_ln = (c.options & VTQ_OPT_FUSED_LAYERNORM) != 0;
    if (e->fuse_ln && !(c.hidden_size == 768 && !e->fp8 && e->lin.terms == 3 && c.num_adapters == 0)) {
        delete e;
        return fail("VTQ_OPT_FUSED_LAYERNORM needs hidden_size 768, a 3-term precision (fp16x3 / bf16x3) and no adapters");
    }
    e->s_ln1.assign(c.num_layers, kSLn); e->s_att.assign(c.num_layers, kSAtt); e->s_ln2.assign(c.num_layers, kSLn); e->s_gelu.assign(c.num_layers, kSGelu);
    e->fp8_static = (c.options & VTQ_OPT_FP8_STATIC_SCALES) != 0;
    if (hipMalloc((void**)&e->err_flag, 16) != hipSuccess || hipMemset(e->err_flag, 0, 16) != hipSuccess) {
        vtq_destroy(e);
        return fail("vtq_create: device allocation failed");
    }
    e->allocs.push_back(e->err_flag);
    e->amax_slot = (float*)(e->err_flag + 2);           // same 16-byte allocation
    if (hipHostMalloc((void**)&e->err_host, 64, hipHostMallocDefault) != hipSuccess) {
        e->err_host = nullptr;
        vtq_destroy(e);
        return fail("vtq_create: pinned host allocation failed");
    }
    if (build(e)) { vtq_destroy(e); return 1; }
    *out = e;
    return 0;
}

void vtq_destroy(vtq_handle e) {
    if (!e) return;
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->err_host) (void)hipHostFree(e->err_host);
    for (void* p : e->ws_allocs) (void)hipFree(p);
    for (auto& ev : e->ev_used) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : e->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete e;
}

int vtq_load_weights(vtq_handle e, const vtq_tensor_desc* descs, int32_t n, void* stream) {
    if (!e || !descs) return fail("vtq_load_weights: null argument");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;                                  // one buffer for the scaled copies, reused in stream order
    int64_t scratch_n = 0;
    struct Free { float*& p; ~Free() { if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); } } } free_scratch{scratch};
    for (int i = 0; i < n; ++i) {
        const vtq_tensor_desc& d = descs[i];
        if (!d.name || !d.data) return fail("vtq_load_weights: descriptor %d has a null field", i);
        auto it = e->slots.find(d.name);
        if (it == e->slots.end()) return fail("vtq_load_weights: unexpected tensor '%s'", d.name);
        Slot& sl = it->second;
        if (sl.numel != d.numel) return fail("vtq_load_weights: '%s' has %lld elements, expected %lld", d.name, (long long)d.numel, (long long)sl.numel);
        if (sl.ignore) { sl.loaded = true; continue; }
        const float* src = (const float*)d.data;
        if (sl.mul != 1.0f) {                                   // scaled copy first (query projection, 3-term attention)
            if (scratch_n < sl.numel) {
                if (scratch) (void)hipFree(scratch);
                scratch = nullptr;
                HIP_TRY(hipMalloc((void**)&scratch, (size_t)sl.numel * 4));
                scratch_n = sl.numel;
            }
            HIP_TRY(launch_scale_copy(src, scratch, sl.numel, sl.mul, s));
            src = scratch;
        }
        if (sl.split && e->fp8) HIP_TRY(launch_quant_rows_fp8(src, sl.dst, sl.scale, (int)(sl.numel / sl.K), (int)sl.K, s, (int)sl.Kp));
        else if (sl.split && sl.Kp != sl.K)
            HIP_TRY(launch_split_rows_pad(src, sl.dst, sl.plane, (int)(sl.numel / sl.K), (int)sl.K, (int)sl.Kp, e->f16, e->wpl, s));
        else if (sl.split) HIP_TRY(launch_split(src, sl.dst, sl.plane, sl.numel, e->f16, e->wpl, s));
        else HIP_TRY(hipMemcpyAsync(sl.dst, src, (size_t)sl.numel * 4, hipMemcpyDeviceToDevice, s));
        sl.loaded = true;
    }
    if (pack_head(e, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    // fp8 mode: scales calibrated on the previous weights do not fit these (activations would clamp at +-448): the next forward
    // calibrates again -- unless the caller installed the scales itself (a checkpoint that carries them; vtq_fp8_set_scales)
    if (e->fp8 && !e->fp8_installed) e->fp8_calibrated = false;
    return 0;
}

size_t vtq_workspace_bytes(vtq_handle e, int32_t B, int32_t N) { return e ? workspace_bytes(e, B, N) : 0; }

int vtq_reserve(vtq_handle e, int32_t B, int32_t N) {
    if (!e || B < 1 || N < 1) return fail("vtq_reserve: bad argument");
    return reserve(e, B, N);
}

int vtq_set_token_trace(vtq_handle e, float* buf) {
    if (!e) return fail("null handle");
    e->trace = buf;
    return 0;
}

int vtq_set_iqa_token(vtq_handle e, int32_t token) {
    if (!e) return fail("null handle");
    if (token < 0 || token > e->cfg.num_extra_tokens)
        return fail("vtq_set_iqa_token: token %d outside [0, %d] (CLS + num_extra_tokens register tokens)", (int)token, (int)e->cfg.num_extra_tokens);
    e->iqa_token = token;
    return 0;
}

int vtq_debug_stop_after(vtq_handle e, int32_t stage) {
    if (!e) return fail("null handle");
    e->dbg_stop = stage;
    return 0;
}

int vtq_debug_buffers(vtq_handle e, void** x, void** lnbuf, void** big, int64_t* rows) {
    if (!e || !e->x) return fail("vtq_debug_buffers: no workspace yet (run a forward first)");
    if (x) *x = e->x;
    if (lnbuf) *lnbuf = e->lnbuf;
    if (big) *big = e->big;
    if (rows) *rows = e->ln_plane / e->H;
    return 0;
}

int vtq_debug_attention_variant(int32_t v) {
    attention_set_variant(v);
    return 0;
}

int vtq_debug_gemm_variant(int32_t v) {
    if (v < GEMM_TILE_AUTO || v > 31) return fail("vtq_debug_gemm_variant: %d", v);
    gemm_set_variant(v);
    return 0;
}

int vtq_debug_cu_partition(int32_t gemm_cus_per_xcd, int32_t attention_cus) {
    if (gemm_cus_per_xcd < 0 || gemm_cus_per_xcd > 32 || attention_cus < 0 || attention_cus > 4096) return fail("vtq_debug_cu_partition: %d, %d", (int)gemm_cus_per_xcd, (int)attention_cus);
    gemm_set_cus_per_xcd(gemm_cus_per_xcd);
    attention_set_cus(attention_cus);
    return 0;
}

int vtq_debug_attention_map(int32_t m) {
    if (m < 0 || m > 1) return fail("vtq_debug_attention_map: %d", (int)m);
    attention_set_map(m);
    return 0;
}

int vtq_debug_cu_map(uint32_t* out, int32_t nblocks, int32_t spin_us, void* stream) {
    if (!out || nblocks < 1 || nblocks > 4096 || spin_us < 0 || spin_us > 100000) return fail("vtq_debug_cu_map: bad argument");
    HIP_TRY(launch_cu_map(out, nblocks, spin_us, (hipStream_t)stream));
    return 0;
}

int vtq_k_gemm_tile_rule(int32_t M, int32_t N, int32_t K, int32_t num) {
    const Num nm = num_from_code(num);
    if (!num_valid(nm)) return -1;
    return gemm_tile_rule(M, N, K, nm);
}

int vtq_k_attention_rule(int32_t nseq, int32_t S_pad, int32_t H, int32_t num, int32_t cus) {
    const Num nm = num_from_code(num);
    if (!num_valid(nm)) return -1;
    return attention_rule(nseq, S_pad, H, nm.terms, cus);
}

int vtq_debug_gemm_diag(void* buf, int32_t shadow) {
    gemm_set_diag((unsigned long long*)buf, shadow);
    return gemm_is_diag_build() ? 1 : 0;
}

int vtq_debug_mfma_stream(int32_t f16, int32_t data, double warm_s, double timed_s, double* tflops, double* ghz, void* stream) {
    if (f16 < 0 || f16 > 1 || data < 0 || data > 2 || !(warm_s >= 0.0) || !(timed_s > 0.0) || warm_s > 30.0 || timed_s > 30.0 || !tflops)
        return fail("vtq_debug_mfma_stream: bad argument");
    HIP_TRY(mfma_stream_measure(f16, data, warm_s, timed_s, tflops, ghz, (hipStream_t)stream));
    return 0;
}

int vtq_profile_enable(vtq_handle e, uint32_t mask) {
    if (!e) return fail("null handle");
    e->prof_mask = mask;
    return 0;
}

int vtq_profile_collect(vtq_handle e, double* ms_sum, int64_t* launches) {
    if (!e || !ms_sum || !launches) return fail("vtq_profile_collect: null argument");
    for (int k = 0; k < VTQ_K_COUNT; ++k) { ms_sum[k] = 0; launches[k] = 0; }
    for (auto& ev : e->ev_used) {
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        ms_sum[ev.cls] += ms;
        launches[ev.cls] += 1;
        e->ev_free.push_back({ev.a, ev.b});
    }
    e->ev_used.clear();
    return 0;
}

// nimg = 2: q_out[B] for (ref, dist); nimg = 3: q_out[2B] = scores of (ref, dist1) then (ref, dist2) with ref encoded once
static int forward_impl(vtq_handle e, int nimg, const float* const* patches, const float* const* pos, const float* const* scales,
                        int32_t B, int32_t N, float* q_out, void* stream, bool tokens_in = false) {
    if (!e) return fail("vtq_forward: null handle");
    if (tokens_in && e->fp8) return fail("vtq_forward_tokens: the fp8 experiment has no pre-embedded input path");
    for (int k = 0; k < nimg; ++k)
        if (!patches[k] || !pos[k]) return fail("vtq_forward: null tensor");
    if (!q_out) return fail("vtq_forward: null output");
    if (B < 1 || N < 1) return fail("vtq_forward: B=%d N=%d", B, N);
    const vtq_config& c = e->cfg;
    const bool use_scales = c.num_scales > 1;
    if (use_scales)
        for (int k = 0; k < nimg; ++k)
            if (!scales[k]) return fail("Model uses scale embedding but scales is passed as None.");   // transformer.py:547-548
    for (auto& kv : e->slots)
        if (!kv.second.loaded) return fail("vtq_forward: weight '%s' was never loaded", kv.first.c_str());
    if (reserve(e, (nimg * B + 1) / 2, N)) return 1;       // capacity is kept in units of sequence pairs
    // fp8 mode: the first forward of an engine calibrates the activation scales on its own batch (see kSPatch); it synchronises
    // the stream once per quantisation point and returns this batch's scores computed with the final scales
    struct CalibGuard { vtq_engine* e; ~CalibGuard() { e->calibrating = false; } } calib_guard{e};
    if (e->fp8 && !e->fp8_static && !e->fp8_calibrated && e->dbg_stop < 0) e->calibrating = true;
    const int ndist = nimg - 1, HB = ndist * B;            // head batch
    hipStream_t s = (hipStream_t)stream;
    const Geometry g = geometry(e, B, N, nimg);
    const int H = e->H, T = e->T;
    {   // tile schedules of this geometry's GEMM shapes: built and uploaded here (first forward of a shape only), not inside a launch
        const int wpl = e->wpl, M = (int)g.M_pad;
        HIP_TRY(gemm_prepare((int)g.P_pad, H, (int)e->PDp, wpl, s));
        HIP_TRY(gemm_prepare(M, 3 * H, H, wpl, s));
        HIP_TRY(gemm_prepare(M, 2 * H, H, wpl, s));
        HIP_TRY(gemm_prepare(M, H, H, wpl, s));
        HIP_TRY(gemm_prepare(M, e->Mdim, H, wpl, s));
        HIP_TRY(gemm_prepare(M, H, e->Mdim, wpl, s));
        if (c.num_adapters > 0) { HIP_TRY(gemm_prepare(M, (int)e->Hqp, H, wpl, s)); HIP_TRY(gemm_prepare(M, H, (int)e->Hqp, wpl, s)); }
    }

    // ---- embeddings (transformer.py:526-562) -------------------------------------------------------------------
    if (tokens_in) {
        // pre-embedded input (transformer.py:534-535): `patches` are (B, N, H) feature rows; no patch convolution
        Prof p(e, s, VTQ_K_CONVERT);
        HIP_TRY(launch_embed_index(pos, use_scales ? scales : nullptr, nimg, e->pidx, e->sidx, e->row_map, B, N, (int)g.P_pad, g.sm, T,
                                   c.pos_grid, c.num_scales, e->err_flag, s));
        HIP_TRY(launch_zero_pad_rows(e->x, g.nseq, g.S, g.sm, H, (int)g.rows_alloc, s));
        HIP_TRY(launch_tokens(e->x, e->cls, e->pos_table, e->extra, g.nseq, g.sm, T, H, s));
        HIP_TRY(launch_embed_rows(patches, nimg, B * N, e->row_map, e->pidx, e->sidx, e->pos_table, use_scales ? e->scale_table : nullptr, e->x, H, s));
    } else {
    {
        Prof p(e, s, VTQ_K_CONVERT);
        if (e->fp8) {
            if (fp8_stage(e, s, e->s_patch, [&](float sc, Fp8Obs ob) { HIP_TRY(launch_pack_patches(patches, nimg, e->big, e->big_plane, B * N, c.patch_dim, (int)g.P_pad, 2, 1, s, sc, (int)e->PDp, ob)); return 0; })) return 1;
        } else HIP_TRY(launch_pack_patches(patches, nimg, e->big, e->big_plane, B * N, c.patch_dim, (int)g.P_pad, e->f16, e->apl, s, 1.0f, (int)e->PDp));
        HIP_TRY(launch_embed_index(pos, use_scales ? scales : nullptr, nimg, e->pidx, e->sidx, e->row_map, B, N, (int)g.P_pad, g.sm, T,
                                   c.pos_grid, c.num_scales, e->err_flag, s));
        HIP_TRY(launch_zero_pad_rows(e->x, g.nseq, g.S, g.sm, H, (int)g.rows_alloc, s));
        HIP_TRY(launch_tokens(e->x, e->cls, e->pos_table, e->extra, g.nseq, g.sm, T, H, s));
    }
    {
        Prof p(e, s, VTQ_K_PATCH);
        GemmArgs a{};
        a.A = e->big; a.a_plane = e->big_plane; a.lda = (int)e->PDp;
        a.W = e->wpatch; a.w_plane = e->ppatch;
        a.M = (int)g.P_pad; a.N = H; a.K = (int)e->PDp;
        a.bias = e->bpatch; a.x = e->x;
        a.row_map = e->row_map; a.idx1 = e->pidx; a.table1 = e->pos_table;
        a.idx2 = e->sidx; a.table2 = use_scales ? e->scale_table : nullptr;
        a.wscale = e->spatch; a.ascale_inv = 1.0f / e->s_patch;
        HIP_TRY(launch_gemm(a, e->lin, EPI_EMBED, s));
    }
    }
    if (e->trace) HIP_TRY(launch_copy_tokens(e->x, e->trace, g.nseq, g.sm, T, H, s));

    // The last 64-key tile of the last sequence reads K / V rows up to 63 past M_pad in the QKV layout of `big`.  Masked keys
    // multiply by probability 0, which only holds for FINITE stale values: a previous forward that overflowed (inf / NaN, see
    // vtq_input_errors bit 1) with a larger batch would otherwise poison this one's first layer.  From layer 1 on the region
    // holds this forward's own fc1 output.
    for (int pl = 0; pl < e->apl; ++pl)
        HIP_TRY(hipMemsetAsync((char*)e->big + ((size_t)pl * e->big_plane + (size_t)g.M_pad * 3 * H) * 2, 0, (size_t)128 * 3 * H * 2, s));

    // ---- encoder (transformer.py:363-378, 275-285) -------------------------------------------------------------
    // the trace tap needs every token row of the last layer; sequences longer than the CLS kernel's LDS score buffer run the
    // full last layer instead (same result)
    // (fp8 mode runs the full last layer: its CLS row then goes through the same e4m3 GEMMs as every other row)
    const bool prune = e->cls_prune && !e->trace && !e->fp8 && c.num_adapters == 0 && g.S <= cls_attention_max_seq();
    if (run_encoder(e, g, s, prune)) return 1;

    // ---- head (vtamiq.py:104-117) ------------------------------------------------------------------------------
    {
        Prof p(e, s, VTQ_K_HEAD);
        float* d = e->hb[0];
        const PlaneOut hp{e->hp[0], e->hp_plane, H, 1, 2, head_first_slope(e)};      // the first head stage's input planes
        if (prune) HIP_TRY(launch_final_diff(e->xcls, e->encw, e->encb, c.diff_scale ? e->diff_gamma : nullptr, d, B, ndist, SeqMap{1, g.nseq, 0}, H, hp, s, e->err_flag));
        else HIP_TRY(launch_final_diff(e->x + (int64_t)e->iqa_token * H, e->encw, e->encb, c.diff_scale ? e->diff_gamma : nullptr, d, B, ndist, g.sm, H, hp, s, e->err_flag));
        if (run_head(e, d, HB, q_out, s, true)) return 1;
    }
    if (e->calibrating) e->fp8_calibrated = true;
    return 0;
}

#ifdef VTQ_WITH_FP8
int vtq_fp8_calibrate(vtq_handle e, const float* patches_ref, const float* patches_dist, const float* pos_ref, const float* pos_dist,
                      const float* scales_ref, const float* scales_dist, int32_t B, int32_t N, float* q_out, void* stream) {
    if (!e || !e->fp8) return fail("vtq_fp8_calibrate: not an fp8 engine");
    e->fp8_calibrated = false;
    e->fp8_installed = false;
    const bool was_static = e->fp8_static;
    e->fp8_static = false;
    const int rc = vtq_forward(e, patches_ref, patches_dist, pos_ref, pos_dist, scales_ref, scales_dist, B, N, q_out, stream);
    e->fp8_static = was_static;
    return rc;
}

int vtq_fp8_get_scales(vtq_handle e, float* out, int32_t cap) {
    if (!e || !e->fp8) return -1;
    const int L = e->cfg.num_layers, n = 1 + 4 * L;
    if (out) {
        std::vector<float> v(n);
        v[0] = e->s_patch;
        for (int i = 0; i < L; ++i) { v[1 + 4 * i] = e->s_ln1[i]; v[2 + 4 * i] = e->s_att[i]; v[3 + 4 * i] = e->s_ln2[i]; v[4 + 4 * i] = e->s_gelu[i]; }
        for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    }
    return n;
}

int vtq_fp8_set_scales(vtq_handle e, const float* in, int32_t n) {
    if (!e || !e->fp8 || !in) return fail("vtq_fp8_set_scales: not an fp8 engine / null argument");
    const int L = e->cfg.num_layers;
    if (n != 1 + 4 * L) return fail("vtq_fp8_set_scales: %d values, expected %d (patches, then LN1 / attention / LN2 / GELU per layer)", n, 1 + 4 * L);
    for (int i = 0; i < n; ++i) {
        int ex = 0;
        if (!(in[i] > 0.0f) || !std::isfinite(in[i]) || frexpf(in[i], &ex) != 0.5f) return fail("vtq_fp8_set_scales: scale %d = %g is not a positive power of two", i, in[i]);
    }
    e->s_patch = in[0];
    for (int i = 0; i < L; ++i) { e->s_ln1[i] = in[1 + 4 * i]; e->s_att[i] = in[2 + 4 * i]; e->s_ln2[i] = in[3 + 4 * i]; e->s_gelu[i] = in[4 + 4 * i]; }
    e->fp8_calibrated = true;
    e->fp8_installed = true;
    return 0;
}

int vtq_fp8_reset(vtq_handle e) {
    if (!e || !e->fp8) return fail("vtq_fp8_reset: not an fp8 engine");
    e->fp8_calibrated = false;                 // the next forward calibrates on its own batch again
    e->fp8_installed = false;
    return 0;
}
#endif

int vtq_forward(vtq_handle e, const float* patches_ref, const float* patches_dist, const float* pos_ref, const float* pos_dist,
                const float* scales_ref, const float* scales_dist, int32_t B, int32_t N, float* q_out, void* stream) {
    const float* p[2] = {patches_ref, patches_dist};
    const float* ps[2] = {pos_ref, pos_dist};
    const float* sc[2] = {scales_ref, scales_dist};
    return forward_impl(e, 2, p, ps, sc, B, N, q_out, stream);
}

int vtq_forward_tokens(vtq_handle e, const float* feats_ref, const float* feats_dist, const float* pos_ref, const float* pos_dist,
                       const float* scales_ref, const float* scales_dist, int32_t B, int32_t N, float* q_out, void* stream) {
    const float* p[2] = {feats_ref, feats_dist};
    const float* ps[2] = {pos_ref, pos_dist};
    const float* sc[2] = {scales_ref, scales_dist};
    return forward_impl(e, 2, p, ps, sc, B, N, q_out, stream, true);
}

int vtq_forward_pairwise(vtq_handle e, const float* const* patches, const float* const* pos, const float* const* scales, int32_t B,
                         int32_t N, float* q_out, void* stream) {
    if (!patches || !pos) return fail("vtq_forward_pairwise: null argument");
    const float* sc[3] = {scales ? scales[0] : nullptr, scales ? scales[1] : nullptr, scales ? scales[2] : nullptr};
    return forward_impl(e, 3, patches, pos, sc, B, N, q_out, stream);
}

int vtq_forward_pairwise_tokens(vtq_handle e, const float* const* feats, const float* const* pos, const float* const* scales, int32_t B,
                                int32_t N, float* q_out, void* stream) {
    if (!feats || !pos) return fail("vtq_forward_pairwise_tokens: null argument");
    const float* sc[3] = {scales ? scales[0] : nullptr, scales ? scales[1] : nullptr, scales ? scales[2] : nullptr};
    return forward_impl(e, 3, feats, pos, sc, B, N, q_out, stream, true);
}

int vtq_input_errors(vtq_handle e, int32_t* flags, void* stream) {
    if (!e || !flags) return fail("vtq_input_errors: null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(e->err_host, e->err_flag, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *flags = *e->err_host;
    HIP_TRY(hipMemsetAsync(e->err_flag, 0, 4, s));      // after the wait: it runs under the caller's next host-side work, ahead of the next forward in stream order
    return 0;
}

// ---- per-kernel entry points -------------------------------------------------------------------------------------
int vtq_k_split(const float* src, void* dst, int64_t plane_stride, int64_t numel, int32_t f16, int32_t planes, void* stream) {
    HIP_TRY(launch_split(src, dst, plane_stride, numel, f16, planes, (hipStream_t)stream));
    return 0;
}

int vtq_k_gemm_schedule(int32_t M, int32_t N, int32_t K, int32_t wplanes, int32_t* out, int32_t cap) {
    if (M < 256 || N < 256 || M % 256 || N % 256 || K < 1 || wplanes < 1 || wplanes > 2) return -1;
    const std::vector<int> v = gemm_tile_schedule(M / 256, N / 256, K, wplanes);
    if (out)
        for (size_t i = 0; i < v.size() && (int32_t)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

int vtq_k_gemm(const void* A, int64_t a_plane, int32_t lda, const void* W, int64_t w_plane, int32_t M, int32_t N, int32_t K,
               int32_t num, int32_t epilogue, const float* bias, const float* gamma, float* x_f32, void* out16,
               int64_t o_plane, int32_t ldo, void* stream) {
    if (epilogue < 0 || epilogue > 2) return fail("vtq_k_gemm: epilogue %d", epilogue);
    const Num nm = num_from_code(num);
    if (!num_valid(nm)) return fail("vtq_k_gemm: operand format code %d", num);
    GemmArgs a{};
    a.A = A; a.a_plane = a_plane; a.lda = lda; a.W = W; a.w_plane = w_plane; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.gamma = gamma; a.x = x_f32; a.out = out16; a.o_plane = o_plane; a.ldo = ldo;
    HIP_TRY(launch_gemm(a, nm, epilogue, (hipStream_t)stream));
    return 0;
}

int vtq_k_gemm_rowln(const void* A, int64_t a_plane, int32_t lda, const void* W, int64_t w_plane, int32_t M, int32_t K, int32_t num,
                     const float* bias, const float* gamma, float* x_f32, const float* ln_w, const float* ln_b, void* out16, int64_t o_plane,
                     void* stream) {
    const Num nm = num_from_code(num);
    if (!num_valid(nm) || nm.terms != 3) return fail("vtq_k_gemm_rowln: operand format code %d (3-term formats only)", num);
    RowLnArgs a{};
    a.A = A; a.a_plane = a_plane; a.lda = lda; a.W = W; a.w_plane = w_plane; a.M = M; a.N = 768; a.K = K; a.bias = bias; a.gamma = gamma;
    a.x = x_f32; a.ln_w = ln_w; a.ln_b = ln_b; a.out = out16; a.o_plane = o_plane;
    HIP_TRY(launch_gemm_rowln(a, nm, (hipStream_t)stream));
    return 0;
}

#ifdef VTQ_WITH_FP8
int vtq_k_quant_rows_fp8(const float* W, void* dst, float* inv_scale, int32_t N, int32_t K, void* stream) {
    if (!W || !dst || !inv_scale || N < 1 || K < 4) return fail("vtq_k_quant_rows_fp8: bad argument");
    HIP_TRY(launch_quant_rows_fp8(W, dst, inv_scale, N, K, (hipStream_t)stream));
    return 0;
}

int vtq_k_quant_fp8(const float* src, void* dst, int64_t numel, float scale, void* stream) {
    HIP_TRY(launch_split(src, dst, 0, numel, 2, 1, (hipStream_t)stream, scale));
    return 0;
}

int vtq_k_gemm_fp8(const void* A8, int32_t lda, const void* W8, const float* wscale, float ascale_inv, int32_t M, int32_t N, int32_t K,
                   int32_t epilogue, const float* bias, const float* gamma, float* x_f32, void* out, int64_t o_plane, int32_t ldo,
                   float out_scale, void* stream) {
    if (epilogue < 0 || epilogue > 2) return fail("vtq_k_gemm_fp8: epilogue %d", epilogue);
    GemmArgs a{};
    a.A = A8; a.lda = lda; a.W = W8; a.M = M; a.N = N; a.K = K; a.bias = bias; a.gamma = gamma; a.x = x_f32; a.out = out; a.o_plane = o_plane;
    a.ldo = ldo; a.wscale = wscale; a.ascale_inv = ascale_inv; a.out_scale = out_scale;
    HIP_TRY(launch_gemm(a, Num{2, 1}, epilogue, (hipStream_t)stream));
    return 0;
}
#endif

int vtq_k_layernorm(const float* x, const float* w, const float* b, void* out, int64_t o_plane, int32_t rows, int32_t H,
                    int32_t f16, int32_t planes, void* stream) {
    HIP_TRY(launch_layernorm(x, w, b, out, o_plane, rows, H, f16, planes, (hipStream_t)stream));
    return 0;
}

int vtq_k_attention(const void* qkv, int64_t plane, void* out, int64_t o_plane, int32_t nseq, int32_t S, int32_t S_pad,
                    int32_t H, int32_t num, void* stream) {
    const Num nm = num_from_code(num);
    if (!num_valid(nm) || nm.terms == 2) return fail("vtq_k_attention: operand format code %d", num);
    HIP_TRY(launch_attention(qkv, plane, out, o_plane, nseq, S, S_pad, H, nm, (hipStream_t)stream));
    return 0;
}

int vtq_k_skinny_linear(const void* xa, int64_t xa_plane, int32_t ldx, const void* W, int64_t w_plane, int32_t R, int32_t N, int32_t K,
                        int32_t num, int32_t epi, const float* bias, const float* post_slope, const float* gamma, const float* res,
                        const float* aux, int32_t ldr, int32_t nsplit, float* y, int32_t ldy, int32_t ycols, void* ya, int64_t ya_plane,
                        int32_t ldya, int32_t pcol0, const float* next_slope, void* stream) {
    const Num nm = num_from_code(num);
    if (!num_valid(nm)) return fail("vtq_k_skinny_linear: operand format code %d", num);
    if (epi < SK_PLAIN || epi > SK_CONVCAT) return fail("vtq_k_skinny_linear: epilogue %d", epi);
    SkinnyArgs a{};
    a.xa = xa; a.xa_plane = xa_plane; a.ldx = ldx; a.W = W; a.w_plane = w_plane; a.R = R; a.N = N; a.K = K; a.bias = bias; a.epi = epi;
    a.post_slope = post_slope; a.gamma = gamma; a.res = res; a.aux = aux; a.ldr = ldr; a.nsplit = nsplit;
    a.y = y; a.ldy = ldy; a.ycols = ycols; a.ya = ya; a.ya_plane = ya_plane; a.ldya = ldya; a.ya_planes = nm.apl(); a.pcol0 = pcol0;
    a.next_slope = next_slope;
    HIP_TRY(launch_skinny(a, nm, (hipStream_t)stream));
    return 0;
}

int vtq_k_diffnet_head(vtq_handle e, const float* d, int32_t HB, float* q_out, void* stream) {
    if (!e || !d || !q_out || HB < 1) return fail("vtq_k_diffnet_head: bad argument");
    for (auto& kv : e->slots)
        if (!kv.second.loaded) return fail("vtq_k_diffnet_head: weight '%s' was never loaded", kv.first.c_str());
    if (reserve(e, (HB + 1) / 2, 1)) return 1;
    return run_head(e, d, HB, q_out, (hipStream_t)stream);
}

int vtq_k_repeat_mean(const float* q, double* out, int32_t R, int32_t N, void* stream) {
    if (!q || !out || R < 1 || N < 1) return fail("vtq_k_repeat_mean: bad argument");
    HIP_TRY(launch_repeat_mean(q, out, R, N, (hipStream_t)stream));
    return 0;
}

int vtq_k_rank_metrics(const double* a, const double* b, int32_t N, int32_t normalize, double* work, int64_t* counts, double* out,
                       void* stream) {
    if (!a || !b || !work || !counts || !out || N < 2) return fail("vtq_k_rank_metrics: bad argument");
    HIP_TRY(launch_rank_metrics(a, b, N, normalize, work, work + N, work + 2 * (int64_t)N, work + 3 * (int64_t)N, (long long*)counts, out,
                                (hipStream_t)stream));
    return 0;
}

int vtq_k_image_normalize(const uint8_t* images, float* out, int32_t NI, int32_t H, int32_t W, const int32_t* flips, const float* mean,
                          const float* std_, void* stream) {
    if (!images || !out || !mean || !std_ || NI < 1 || H < 1 || W < 1) return fail("vtq_k_image_normalize: bad argument");
    HIP_TRY(launch_image_normalize(images, out, NI, H, W, flips, mean, std_, (hipStream_t)stream));
    return 0;
}

int vtq_k_avgpool2(const float* in, float* out, int32_t NC, int32_t H, int32_t W, void* stream) {
    HIP_TRY(launch_avgpool2(in, out, NC, H, W, (hipStream_t)stream));
    return 0;
}

int vtq_k_gather_patches(const float* const* levels, const int32_t* hs, const int32_t* ws, int32_t nlevels, const int32_t* samples,
                         const int32_t* scale_ids, float* patches, float* pos, float* scales, int32_t NI, int32_t N, int32_t patch_size,
                         void* stream) {
    if (!levels || !hs || !ws || !samples || !patches || !pos) return fail("vtq_k_gather_patches: null argument");
    if (patch_size != 16 && patch_size != 8) return fail("vtq_k_gather_patches: patch_size %d (16 or 8)", patch_size);
    HIP_TRY(launch_gather_patches(levels, hs, ws, nlevels, samples, scale_ids, patches, pos, scales, NI, N, (hipStream_t)stream, patch_size));
    return 0;
}

}  // extern "C"
