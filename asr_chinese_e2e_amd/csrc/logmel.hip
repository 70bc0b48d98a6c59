// Log-mel front end on device: framing (centred, reflect padding) -> Hann window -> 400-point
// real DFT power spectrum -> mel filterbank -> log, then per-utterance scalar normalisation and
// low-frame-rate stacking.  Replaces the CPU/torchaudio path of
// Predictor/data_handler/processor.py:33-46, 74-100.
#include "asr_common.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = NFFT / 2 + 1;

// One workgroup = 32 consecutive frames of one utterance; the DFT is a GEMM on the fp32 matrix pipe:
//   C[32 frames][402] = X[32][400] (windowed frames, LDS)  x  D[400][402],
//   D[n][c] = cos(2 pi n c / 400) for c < 201,  -sin(2 pi n (c-201) / 400) for 201 <= c < 402,
// with D never stored: the lane that owns column c keeps cos / sin of its angle in registers and
// rotates them by 2 theta per step (exact restart every 50 steps).  v_mfma_f32_32x32x2_f32 is exact fp32 multiply-accumulate, so the
// accuracy is that of the direct sum.  Then power = re^2 + im^2 (from C, kept in LDS) times the mel
// filterbank, again on the matrix pipe, log, store.  (The first version gave one DFT bin to each
// thread and one frame to each workgroup: 80 k scalar MACs per frame through LDS reads, 0.6 ms for
// 32 x 5 s of audio = 17 % of a training step.)
constexpr int FR = 32;            // frames per workgroup
constexpr int XS = 401;           // LDS row stride of the frame tile (odd: rows hit distinct banks)
constexpr int NC = 2 * NBIN;      // 402 DFT output columns (re | im)
constexpr int CS = 417;           // LDS row stride of C (13 column tiles of 32 = 416, +1)
constexpr int CT = 13;            // column tiles

__global__ __launch_bounds__(256, 2) void logmel_kernel(const float* __restrict__ wav, const int32_t* __restrict__ wav_len, const float* __restrict__ window,
                                                     const float* __restrict__ melfb, float* __restrict__ feat, int Smax, int Tmax, int n_mels) {
    __shared__ float buf[FR * CS];            // frames (32 x 401), later C (32 x 417)
    __shared__ float tw_c[NFFT], tw_s[NFFT];
    const int t0 = blockIdx.x * FR, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int len = wav_len[b];
    const int Tb = len > 0 ? min(1 + len / HOP, Tmax) : 0;
    float* out = feat + ((size_t)b * Tmax + t0) * n_mels;
    const int rows_here = min(FR, Tmax - t0);
    if (t0 >= Tb) {                            // nothing but padding: zeros
        for (int i = tid; i < rows_here * n_mels; i += 256) out[i] = 0.f;
        return;
    }
    const float* wv = wav + (size_t)b * Smax;
    for (int i = tid; i < FR * NFFT; i += 256) {
        const int f = i / NFFT, n = i - f * NFFT;
        int idx = (t0 + f) * HOP - NFFT / 2 + n;
        if (idx < 0) idx = -idx;                       // reflect (no edge repeat)
        if (idx >= len) idx = 2 * (len - 1) - idx;
        idx = idx < 0 ? 0 : (idx >= len ? len - 1 : idx);
        buf[f * XS + n] = (t0 + f < Tb) ? wv[idx] * window[n] : 0.f;
    }
    for (int n = tid; n < NFFT; n += 256) {            // exact cos / sin(2 pi n / 400): start values and rotation steps
        float sv, cv;
        sincospif(2.f * (float)n / (float)NFFT, &sv, &cv);
        tw_c[n] = cv;
        tw_s[n] = sv;
    }
    __syncthreads();
    // ---- DFT: wave w owns the four column tiles 4w .. 4w+3 (16 tiles cover 512 >= 402 columns; the
    // three all-zero ones keep the code branch-free and the four waves equally loaded).  The lane
    // that owns column c needs cos / sin(2 pi k bin / 400) for k = kh, kh+2, ...: a rotation by
    // 2 theta per step in registers (a table in LDS would be read at stride k*bin: up to 32-way bank
    // conflicts).  200 steps of the fp32 recurrence drift by ~2e-5 relative: 1e-5 in the log domain.
    const int r = lane & 31, kh = lane >> 5;
    f32x16 acc[4];
    float tc[4], ts[4], rc[4], rs[4], sgn_c[4], sgn_s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        const int c = (4 * w + q) * 32 + r;
        const bool live = c < NC, is_im = c >= NBIN;
        const int bin = is_im ? c - NBIN : c;
        sgn_c[q] = live && !is_im ? 1.f : 0.f;         // operand = sgn_c * cos + sgn_s * sin
        sgn_s[q] = live && is_im ? -1.f : 0.f;
        const int j2 = (2 * bin) % NFFT, j0 = (kh * bin) % NFFT;
        rc[q] = tw_c[j2];                              // rotation by 2 theta
        rs[q] = tw_s[j2];
        tc[q] = tw_c[j0];                              // k = kh
        ts[q] = tw_s[j0];
    }
    const float* xrow = buf + r * XS + kh;
#pragma unroll 4
    for (int kk = 0; kk < NFFT / 2; ++kk) {
        const float a = xrow[2 * kk];                  // X[frame r][k = 2 kk + kh]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = fmaf(sgn_c[q], tc[q], sgn_s[q] * ts[q]);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, acc[q], 0, 0, 0);   // A rows = frames, B cols = DFT columns
            const float nc = fmaf(tc[q], rc[q], -ts[q] * rs[q]);
            ts[q] = fmaf(ts[q], rc[q], tc[q] * rs[q]);
            tc[q] = nc;
        }
    }
    __syncthreads();                                   // everyone is done with the frames: the buffer becomes C
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = (4 * w + q) * 32 + r;
        if (c >= CT * 32) continue;                    // columns past the C row (all zero anyway)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int frame = (i & 3) + 8 * (i >> 2) + 4 * kh;        // accumulator row
            buf[frame * CS + c] = acc[q][i];
        }
    }
    __syncthreads();
    // ---- mel: out[32][n_mels] = power[32][201] x melfb[201][n_mels]; wave w owns mel columns 32 w .. 32 w + 31
    for (int mt = w; mt * 32 < n_mels; mt += 4) {
        const int m = mt * 32 + r;
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
        const float* crow = buf + r * CS;
        // the filterbank column of this lane is fetched a quarter at a time into registers BEFORE the
        // MFMAs that use it: issued one by one between MFMAs, each L2 round trip would be exposed
        constexpr int KSTEPS = (NBIN + 1) / 2, CHUNK = 26;     // 101 steps in 4 chunks
#pragma unroll 1
        for (int k0 = 0; k0 < KSTEPS; k0 += CHUNK) {
            float mbv[CHUNK];
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) {
                const int k = 2 * (k0 + i) + kh;
                mbv[i] = (k0 + i < KSTEPS && k < NBIN && m < n_mels) ? melfb[(size_t)k * n_mels + m] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) {
                const int k = 2 * (k0 + i) + kh;
                float pa = 0.f;
                if (k0 + i < KSTEPS && k < NBIN) {
                    const float re = crow[k], im = crow[NBIN + k];
                    pa = re * re + im * im;
                }
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, mbv[i], o, 0, 0, 0);
            }
        }
        if (m < n_mels) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int frame = (i & 3) + 8 * (i >> 2) + 4 * kh;
                if (frame < rows_here) out[(size_t)frame * n_mels + m] = (t0 + frame < Tb) ? logf(o[i] + 1e-20f) : 0.f;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void utt_norm_lfr_kernel(const float* __restrict__ feat, const int32_t* __restrict__ wav_len, const int32_t* __restrict__ masks,
                                                            T* __restrict__ out, int32_t* __restrict__ out_len, int Tmax, int n_mels, int m, int n, int Tlfr_max) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = wav_len[b];
    const int Tb = len > 0 ? min(1 + len / HOP, Tmax) : 0;
    const int Tl = min((Tb + n - 1) / n, Tlfr_max);
    const float* f = feat + (size_t)b * Tmax * n_mels;
    const int cnt = Tb * n_mels;
    float s = 0.f;
    for (int i = tid; i < cnt; i += 1024) s += f[i];
    const float mean = cnt > 0 ? block_sum(s, red) / (float)cnt : 0.f;
    float q = 0.f;
    for (int i = tid; i < cnt; i += 1024) { const float d = f[i] - mean; q += d * d; }
    const float var = cnt > 1 ? block_sum(q, red) / (float)(cnt - 1) : 1.f;  // unbiased (torch .std())
    const float rstd = rsqrtf(var);
    // SpecAugment of the reference (augments.py:4-42 via processor.py:52-58): a time mask [t0, t1)
    // filled with the mean of the normalised feature, THEN a mel mask [f0, f1) filled with the mean of
    // the time-masked feature (cloned.mean() at each call).  The ranges come from the host's RNG.
    int t0 = 0, t1 = 0, f0 = 0, f1 = 0;
    float fill_t = 0.f, fill_f = 0.f;
    if (masks) {
        t0 = min(max(masks[4 * b + 0], 0), Tb);
        t1 = min(max(masks[4 * b + 1], t0), Tb);
        f0 = min(max(masks[4 * b + 2], 0), n_mels);
        f1 = min(max(masks[4 * b + 3], f0), n_mels);
        float sa = 0.f, sm = 0.f;
        for (int i = tid; i < cnt; i += 1024) {
            const float v = (f[i] - mean) * rstd;
            const int t = i / n_mels;
            sa += v;
            if (t >= t0 && t < t1) sm += v;
        }
        const float sum_all = block_sum(sa, red), sum_masked = block_sum(sm, red);
        fill_t = cnt > 0 ? sum_all / (float)cnt : 0.f;
        fill_f = cnt > 0 ? (sum_all - sum_masked + (float)((t1 - t0) * n_mels) * fill_t) / (float)cnt : 0.f;
    }
    const int W = m * n_mels;
    T* o = out + (size_t)b * Tlfr_max * W;
    const int total = Tlfr_max * W;
    for (int i = tid; i < total; i += 1024) {
        const int r = i / W, c = i - r * W;
        float val = 0.f;
        if (r < Tl) {
            const int j = c / n_mels, mm = c - j * n_mels;
            const int src = min(r * n + j, Tb - 1);   // tail frames repeat the last input frame
            val = (f[(size_t)src * n_mels + mm] - mean) * rstd;
            if (mm >= f0 && mm < f1) val = fill_f;
            else if (src >= t0 && src < t1) val = fill_t;
        }
        o[i] = from_f32<T>(val);
    }
    if (tid == 0) out_len[b] = Tl;
}

}  // namespace

extern "C" int asr_logmel_fwd(const float* wav, const int32_t* wav_len, const float* window, const float* melfb, float* feat, int B, int Smax, int Tmax,
                              int n_mels, void* stream) {
    if (!wav || !wav_len || !window || !melfb || !feat) ASR_FAIL(ASR_EINVAL, "asr_logmel_fwd: null pointer");
    if (B <= 0 || Smax <= NFFT / 2 || Tmax <= 0 || n_mels <= 0 || B > 65535) ASR_FAIL(ASR_EINVAL, "asr_logmel_fwd: bad shape B=%d Smax=%d Tmax=%d n_mels=%d", B, Smax, Tmax, n_mels);
    dim3 grid(ceil_div(Tmax, FR), B);
    logmel_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(wav, wav_len, window, melfb, feat, Smax, Tmax, n_mels);
    ASR_CHECK_LAUNCH("asr_logmel_fwd");
    return ASR_OK;
}

extern "C" int asr_utt_norm_augment_lfr_fwd(const float* feat, const int32_t* wav_len, const int32_t* masks, void* out, int32_t* out_len, int B, int Tmax,
                                            int n_mels, int m, int n, int Tlfr_max, int dtype, void* stream) {
    if (!feat || !wav_len || !out || !out_len) ASR_FAIL(ASR_EINVAL, "asr_utt_norm_augment_lfr_fwd: null pointer");
    if (B <= 0 || Tmax <= 0 || n_mels <= 0 || m <= 0 || n <= 0 || Tlfr_max <= 0) ASR_FAIL(ASR_EINVAL, "asr_utt_norm_augment_lfr_fwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ASR_F32) utt_norm_lfr_kernel<float><<<B, 1024, 0, st>>>(feat, wav_len, masks, (float*)out, out_len, Tmax, n_mels, m, n, Tlfr_max);
    else if (dtype == ASR_BF16) utt_norm_lfr_kernel<bf16_t><<<B, 1024, 0, st>>>(feat, wav_len, masks, (bf16_t*)out, out_len, Tmax, n_mels, m, n, Tlfr_max);
    else ASR_FAIL(ASR_EDTYPE, "asr_utt_norm_augment_lfr_fwd: dtype %d", dtype);
    ASR_CHECK_LAUNCH("asr_utt_norm_augment_lfr_fwd");
    return ASR_OK;
}

extern "C" int asr_utt_norm_lfr_fwd(const float* feat, const int32_t* wav_len, void* out, int32_t* out_len, int B, int Tmax, int n_mels, int m, int n,
                                    int Tlfr_max, int dtype, void* stream) {
    return asr_utt_norm_augment_lfr_fwd(feat, wav_len, nullptr, out, out_len, B, Tmax, n_mels, m, n, Tlfr_max, dtype, stream);
}
