// wav -> normalised dB mel-spectrogram on the device: the input side of generate.py.
// Restates /root/reference/data_load.py:51-54 with audio.py:102-141 (librosa.stft: centred, reflect-padded, hann window of
// win_length zero-padded to n_fft), :232-243 (Slaney mel filterbank, passed in by the host), :327-356 (amplitude_to_db:
// amin 1e-5, top_db 80 below the utterance maximum) and :254-286 (normalise to [-1, 1]).
// A few hundred MFLOP per utterance: one workgroup per frame, a direct real DFT with fp64 accumulation (the dB scale
// turns relative errors of small bins into absolute ones, so the transform is done in the precision the numpy
// restatement uses), twiddles from an LDS table.  Not a hot path; it exists so that generate() on wav input keeps the mel
// on the device.
#include "pwv_common.h"

namespace pwv {

constexpr int kMaxFft = 2048;

// db_raw[n, frame, m] = 10 log10(max(amin^2, mel^2)),  mel = fb[m, :] . |rfft(window * frame)|
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                        const float* __restrict__ fb, float* __restrict__ db, int L, int n_fft,
                                                        int hop, int frames, int n_mels, float amin) {
    extern __shared__ double sm[];             // [n_fft] frame, [n_fft] cos, [n_fft] sin, [n_fft/2 + 1] magnitude
    double* fr = sm;
    double* ct = sm + n_fft;
    double* st = ct + n_fft;
    double* mag = st + n_fft;
    const int f = blockIdx.x, n = blockIdx.y;
    const float* w = wav + (size_t)n * L;
    for (int i = threadIdx.x; i < n_fft; i += 256) {
        int t = f * hop + i - n_fft / 2;       // centred frame; np.pad(mode='reflect'): -k -> k, L-1+k -> L-1-k
        if (t < 0) t = -t;
        if (t >= L) t = 2 * (L - 1) - t;
        t = t < 0 ? 0 : (t >= L ? L - 1 : t);
        fr[i] = (double)w[t] * (double)window[i];
        double s, c;
        sincospi(2.0 * i / n_fft, &s, &c);
        ct[i] = c;
        st[i] = s;
    }
    __syncthreads();
    const int bins = n_fft / 2 + 1;
    for (int b = threadIdx.x; b < bins; b += 256) {
        double re = 0.0, im = 0.0;
        int k = 0;                              // (b * i) mod n_fft
        for (int i = 0; i < n_fft; ++i) {
            re = fma(fr[i], ct[k], re);
            im = fma(fr[i], st[k], im);
            k += b;
            if (k >= n_fft) k -= n_fft;
        }
        mag[b] = sqrt(re * re + im * im);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_mels; m += 256) {
        double acc = 0.0;
        const float* row = fb + (size_t)m * bins;
        for (int b = 0; b < bins; ++b) acc = fma((double)row[b], mag[b], acc);
        const double p = acc * acc, floor = (double)amin * (double)amin;
        db[((size_t)n * frames + f) * n_mels + m] = (float)(10.0 * log10(p > floor ? p : floor));
    }
}

// per utterance: top_db clip against the maximum of the whole spectrogram, then (clip((db - min)/(max - min), 0, 1) - .5) * 2
__global__ __launch_bounds__(1024) void db_normalize_kernel(float* __restrict__ db, int count, float top_db, float max_db, float min_db,
                                                            int normalise) {
    __shared__ float red[1024];
    float* p = db + (size_t)blockIdx.x * count;
    float mx = -3.0e38f;
    for (int i = threadIdx.x; i < count; i += 1024) mx = fmaxf(mx, p[i]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const float lo = red[0] - top_db;
    for (int i = threadIdx.x; i < count; i += 1024) {
        float v = fmaxf(p[i], lo);
        if (normalise) {
            v = (v - min_db) / (max_db - min_db);
            v = (fminf(fmaxf(v, 0.f), 1.f) - 0.5f) * 2.f;
        }
        p[i] = v;
    }
}

}  // namespace pwv

using namespace pwv;

extern "C" {

int pwv_wav_to_mel_db_f32(const float* wav, const float* window, const float* mel_basis, float* mel, int N, int L, int n_fft, int hop,
                          int n_mels, float amin, float top_db, float max_db, float min_db, int normalise, pwv_stream_t stream) {
    PWV_CHECK_ARG(wav && window && mel_basis && mel, "pwv_wav_to_mel_db_f32: NULL pointer");
    PWV_CHECK_ARG(N >= 1 && L >= 2 && hop >= 1 && n_mels >= 1, "pwv_wav_to_mel_db_f32: bad shape");
    PWV_CHECK_ARG(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= kMaxFft && n_fft / 2 < L, "pwv_wav_to_mel_db_f32: n_fft must be even, <= %d and < 2 L", kMaxFft);
    PWV_CHECK_ARG(!normalise || max_db != min_db, "pwv_wav_to_mel_db_f32: max_db == min_db");
    const int frames = 1 + L / hop;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)(3 * n_fft + n_fft / 2 + 1) * sizeof(double);
    hipLaunchKernelGGL(stft_mel_kernel, dim3(frames, N), dim3(256), smem, s, wav, window, mel_basis, mel, L, n_fft, hop, frames, n_mels, amin);
    hipLaunchKernelGGL(db_normalize_kernel, dim3(N), dim3(1024), 0, s, mel, frames * n_mels, top_db, max_db, min_db, normalise);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

}  // extern "C"
