// Log-mel front end on device: framing (centred, reflect padding) -> Hann window -> 400-point
// real DFT power spectrum -> mel filterbank -> log, then per-utterance scalar normalisation and
// low-frame-rate stacking.  Replaces the CPU/torchaudio path of
// Predictor/data_handler/processor.py:33-46, 74-100.  One workgroup per frame: the windowed
// frame and the 400-entry twiddle table sit in LDS; each thread owns one DFT bin.
#include "asr_common.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = NFFT / 2 + 1;

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, const int32_t* __restrict__ wav_len, const float* __restrict__ window,
                                                     const float* __restrict__ melfb, float* __restrict__ feat, int Smax, int Tmax, int n_mels) {
    __shared__ float xs[NFFT];
    __shared__ float tw_c[NFFT], tw_s[NFFT];
    __shared__ float pw[NBIN + 3];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int len = wav_len[b];
    const int Tb = len > 0 ? 1 + len / HOP : 0;
    float* out = feat + ((size_t)b * Tmax + t) * n_mels;
    if (t >= Tb) {
        for (int m = tid; m < n_mels; m += 256) out[m] = 0.f;
        return;
    }
    const float* w = wav + (size_t)b * Smax;
    for (int n = tid; n < NFFT; n += 256) {
        int idx = t * HOP - NFFT / 2 + n;
        if (idx < 0) idx = -idx;                       // reflect (no edge repeat)
        if (idx >= len) idx = 2 * (len - 1) - idx;
        idx = idx < 0 ? 0 : idx;
        xs[n] = w[idx] * window[n];
        float s, c;
        sincospif(2.f * (float)n / (float)NFFT, &s, &c);
        tw_c[n] = c;
        tw_s[n] = s;
    }
    __syncthreads();
    if (tid < NBIN) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < NFFT; ++n) {
            const float x = xs[n];
            re += x * tw_c[idx];
            im -= x * tw_s[idx];
            idx += tid;
            if (idx >= NFFT) idx -= NFFT;
        }
        pw[tid] = re * re + im * im;
    }
    __syncthreads();
    for (int m = tid; m < n_mels; m += 256) {
        float a = 0.f;
        for (int kb = 0; kb < NBIN; ++kb) a += pw[kb] * melfb[(size_t)kb * n_mels + m];
        out[m] = logf(a + 1e-20f);
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
    dim3 grid(Tmax, B);
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
