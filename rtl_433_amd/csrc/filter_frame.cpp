// filter_frame.cpp -- r433_filter_frame / r433_envelope_host: the function-level seam of the reference's
// include/baseband.h on HOST buffers (one frame per call, filter state in and out), staged through device buffers of
// the library's own.  The work itself is phases A + B of the detection kernel (k_wave<.., SEAM>: chunk-parallel exact
// low-passes) and k_envelope; nothing is computed on the host.  One caller at a time (the reference is single-threaded).
#include "host_common.hpp"

using namespace r433;

namespace {

struct SeamCtx {
    DevBuf<uint8_t> d_in;
    DevBuf<int16_t> d_am, d_fm;
    DevBuf<uint16_t> d_env;
    DevBuf<StreamState> d_state;
    DevBuf<int> d_init;
    DevBuf<uint32_t> d_sum;
    DevBuf<uint8_t> d_out;
    hipStream_t st = nullptr;
    int ensure_stream()
    {
        if (!st)
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return 0;
    }
};
SeamCtx g_seam;

} // namespace

extern "C" {

int r433_filter_frame(uint32_t kind, void const *h_in, uint32_t n_samples, int16_t *h_out, r433_filter_carry *carry,
        int32_t a16, int32_t b16, int64_t a32, int64_t b32)
{
    if (kind < R433_FILTER_AM || kind > R433_FILTER_FM_CS16)
        return fail(R433_EINVAL, "unknown filter kind %u", kind);
    if (n_samples == 0)
        return 0; // "Prevent out of bounds access", src/baseband.c:156-158
    if (!h_in || !h_out || !carry)
        return fail(R433_EINVAL, "null argument");
    if (r433_device_count() < 0)
        return R433_ENODEV;
    uint32_t const ss = kind == R433_FILTER_FM_CS16 ? 4 : 2;
    uint64_t const bytes = (uint64_t)n_samples * ss;
    if (bytes > 0xfffffff0ull)
        return fail(R433_EINVAL, "frames are limited to 4 GiB");
    SeamCtx &c = g_seam;
    int rc;
    if ((rc = c.ensure_stream()) || (rc = c.d_in.ensure(bytes + 64)) || (rc = c.d_am.ensure(n_samples + 64)) || (rc = c.d_fm.ensure(n_samples + 64))
            || (rc = c.d_state.ensure(1)) || (rc = c.d_init.ensure(8)))
        return rc;
    int const init[6] = {carry->am_y, carry->am_x, carry->fm_y, carry->fm_x, carry->last_i, carry->last_q};
    HIP_TRY(hipMemcpyAsync(c.d_in.p, h_in, bytes, hipMemcpyHostToDevice, c.st));
    HIP_TRY(hipMemcpyAsync(c.d_init.p, init, sizeof(init), hipMemcpyHostToDevice, c.st));

    StreamParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.iq = c.d_in.p;
    sp.stride_bytes = (bytes + 15) & ~15ull;
    sp.uniform_bytes = (uint32_t)bytes;
    sp.n_streams = 1;
    sp.frame_samples = (n_samples + 63u) & ~63u; // one frame: no frame boundary inside
    sp.flags = RUN_NOFLUSH | (kind == R433_FILTER_AM ? RUN_ENV_RAW16 : 0u);
    sp.enable_fm = kind != R433_FILTER_AM;
    sp.a16 = a16, sp.b16 = b16, sp.a32 = a32, sp.b32 = b32;
    sp.state = c.d_state.p;
    sp.frames_cap = 1;
    sp.tap_am = c.d_am.p;
    sp.tap_fm = c.d_fm.p;
    sp.tap_stride = n_samples;
    sp.seam_init = c.d_init.p;
    launch_filters(sp, ss, c.st);
    HIP_TRY(hipGetLastError());
    StreamState S;
    HIP_TRY(hipMemcpyAsync(h_out, kind == R433_FILTER_AM ? c.d_am.p : c.d_fm.p, (size_t)n_samples * 2, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipMemcpyAsync(&S, c.d_state.p, sizeof(S), hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    if (S.overflow)
        return fail(R433_EHIP, "filter carry could not be proven (code %u)", S.overflow);
    if (kind == R433_FILTER_AM) {
        carry->am_y = S.lpf_y;
        carry->am_x = S.lpf_x;
    }
    else {
        carry->fm_y = S.fm_yf;
        carry->fm_x = S.fm_xf;
        // the last IQ sample, as the discriminator sees it
        if (ss == 2) {
            uint8_t const *q = (uint8_t const *)h_in + (size_t)(n_samples - 1) * 2;
            carry->last_i = (int)q[0] - 128;
            carry->last_q = (int)q[1] - 128;
        }
        else {
            int16_t const *q = (int16_t const *)h_in + (size_t)(n_samples - 1) * 2;
            carry->last_i = q[0];
            carry->last_q = q[1];
        }
    }
    return 0;
}

int r433_envelope_host(uint32_t kind, void const *h_iq, uint16_t *h_env, uint32_t n_samples, uint32_t *sum)
{
    if (kind > R433_ENV_TRUE_CS16)
        return fail(R433_EINVAL, "unknown envelope kind %u", kind);
    if (sum)
        *sum = 0;
    if (n_samples == 0)
        return 0;
    if (!h_iq || !h_env)
        return fail(R433_EINVAL, "null argument");
    if (r433_device_count() < 0)
        return R433_ENODEV;
    uint32_t const ss = (kind == R433_ENV_MAG_CS16 || kind == R433_ENV_TRUE_CS16) ? 4 : 2;
    SeamCtx &c = g_seam;
    int rc;
    if ((rc = c.ensure_stream()) || (rc = c.d_in.ensure((size_t)n_samples * ss + 64)) || (rc = c.d_env.ensure(n_samples + 64)) || (rc = c.d_sum.ensure(4)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c.d_in.p, h_iq, (size_t)n_samples * ss, hipMemcpyHostToDevice, c.st));
    HIP_TRY(hipMemsetAsync(c.d_sum.p, 0, sizeof(uint32_t), c.st));
    launch_envelope((int)kind, c.d_in.p, c.d_env.p, n_samples, c.d_sum.p, c.st);
    HIP_TRY(hipGetLastError());
    uint32_t s = 0;
    HIP_TRY(hipMemcpyAsync(h_env, c.d_env.p, (size_t)n_samples * 2, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipMemcpyAsync(&s, c.d_sum.p, sizeof(s), hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    if (sum)
        *sum = s;
    return 0;
}

// r433_dump_convert for a caller whose frame sits in HOST memory (dropin/r_flow_hip.c: the -w / -W sample dumpers of the file
// loop, one frame per call): staged through device buffers of the library's own, converted by the same kernel.
int r433_dump_convert_host(int format, uint32_t sample_size, void const *h_in, void *h_out, uint64_t n_out)
{
    if (format < R433_DUMP_CU8_IQ || format > R433_DUMP_F32_Q)
        return fail(R433_EINVAL, "unknown dump format %d", format);
    if (sample_size != 2 && sample_size != 4)
        return fail(R433_EINVAL, "sample_size must be 2 (cu8) or 4 (cs16)");
    if (n_out == 0)
        return 0;
    if (!h_in || !h_out)
        return fail(R433_EINVAL, "null argument");
    if (r433_device_count() < 0)
        return R433_ENODEV;
    // bytes in: IQ formats and the I / Q picks read IQ components of the input's own width; the AM / FM formats read an s16 stream
    bool const from_taps = format == R433_DUMP_S16_AM || format == R433_DUMP_S16_FM || format == R433_DUMP_F32_AM || format == R433_DUMP_F32_FM;
    bool const pick = format == R433_DUMP_F32_I || format == R433_DUMP_F32_Q;
    uint64_t const in_bytes = from_taps ? n_out * 2 : pick ? n_out * sample_size : n_out * (sample_size / 2);
    uint64_t const out_width = (format == R433_DUMP_CU8_IQ || format == R433_DUMP_CS8_IQ) ? 1 : (format == R433_DUMP_CS16_IQ || format == R433_DUMP_S16_AM || format == R433_DUMP_S16_FM) ? 2 : 4;
    uint64_t const out_bytes = n_out * out_width;
    if (in_bytes > 0xfffffff0ull || out_bytes > 0xfffffff0ull)
        return fail(R433_EINVAL, "frames are limited to 4 GiB");
    SeamCtx &c = g_seam;
    int rc;
    if ((rc = c.ensure_stream()) || (rc = c.d_in.ensure(in_bytes + 64)) || (rc = c.d_out.ensure(out_bytes + 64)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c.d_in.p, h_in, in_bytes, hipMemcpyHostToDevice, c.st));
    if ((rc = r433_dump_convert(format, sample_size, c.d_in.p, c.d_out.p, n_out, c.st)) < 0)
        return rc;
    HIP_TRY(hipMemcpyAsync(h_out, c.d_out.p, out_bytes, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    return 0;
}

} // extern "C"
