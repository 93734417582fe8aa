// ofl_runtime.hip -- device selection, memory, streams, events (C ABI in include/ofl.h).
// Plain HIP runtime calls; one process drives one GPU.
#include "ofl_common.h"
#include <stdarg.h>
#include <stdlib.h>

namespace ofl {

static thread_local char g_err[512] = "";

Runtime &rt()
{
    static Runtime r;
    return r;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    int code = (e == hipErrorOutOfMemory) ? OFL_E_NOMEM : OFL_E_HIP;
    (void)hipGetLastError();   // clear the sticky error so later calls report their own status
    return fail(code, "%s: %s", what, hipGetErrorString(e));
}

int need_device()
{
    if (!rt().ready) return fail(OFL_E_NODEVICE, "no HIP device selected: call ofl_init(device) first");
    // HIP's current device is per THREAD and defaults to 0: a caller on another thread than the one that ran
    // ofl_init (a Python worker thread, a launcher's helper) would otherwise create streams / communicators on GPU 0
    // while the buffers live on rt().device.  One cheap runtime call per thread.
    static thread_local int bound = -1;
    if (bound != rt().device) {
        hipError_t e = hipSetDevice(rt().device);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
        bound = rt().device;
    }
    return OFL_OK;
}

}  // namespace ofl

using namespace ofl;

extern "C" {

int ofl_abi_version(void) { return OFL_ABI_VERSION; }

const char *ofl_last_error(void) { return g_err; }

int ofl_device_count(int *count)
{
    if (!count) return fail(OFL_E_INVALID, "ofl_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return OFL_OK;
}

int ofl_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(OFL_E_NODEVICE, "ofl_init: no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) return fail(OFL_E_INVALID, "ofl_init: device %d out of range [0,%d)", device, n);
    Runtime &r = rt();
    if (r.ready && r.device == device) return OFL_OK;
    OFL_HIP(hipSetDevice(device));
    if (r.stream) {
        (void)hipStreamDestroy(r.stream);
        r.stream = nullptr;
    }
    OFL_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    OFL_HIP(hipGetDeviceProperties(&prop, device));
    r.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    r.device = device;
    r.ready = true;
    return OFL_OK;
}

int ofl_device_name(char *buf, size_t buflen)
{
    OFL_TRY(need_device());
    if (!buf || buflen == 0) return fail(OFL_E_INVALID, "ofl_device_name: NULL buffer");
    hipDeviceProp_t prop;
    OFL_HIP(hipGetDeviceProperties(&prop, rt().device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return OFL_OK;
}

int ofl_malloc(void **dptr, size_t bytes)
{
    OFL_TRY(need_device());
    if (!dptr) return fail(OFL_E_INVALID, "ofl_malloc: NULL");
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    OFL_HIP(hipMalloc(dptr, bytes));
    return OFL_OK;
}

int ofl_free(void *dptr)
{
    if (!dptr) return OFL_OK;
    OFL_TRY(need_device());
    OFL_HIP(hipFree(dptr));
    return OFL_OK;
}

int ofl_host_alloc(void **hptr, size_t bytes)
{
    OFL_TRY(need_device());
    if (!hptr) return fail(OFL_E_INVALID, "ofl_host_alloc: NULL");
    *hptr = nullptr;
    if (bytes == 0) bytes = 16;
    OFL_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return OFL_OK;
}

int ofl_host_free(void *hptr)
{
    if (!hptr) return OFL_OK;
    OFL_TRY(need_device());
    OFL_HIP(hipHostFree(hptr));
    return OFL_OK;
}

int ofl_download_async(void *host, const void *dptr, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (bytes == 0) return OFL_OK;
    if (!dptr || !host) return fail(OFL_E_INVALID, "ofl_download_async: NULL pointer");
    OFL_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, stream_of(stream)));
    return OFL_OK;
}

int ofl_memset(void *dptr, int value, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (bytes == 0) return OFL_OK;
    OFL_HIP(hipMemsetAsync(dptr, value, bytes, stream_of(stream)));
    return OFL_OK;
}

int ofl_upload(void *dptr, const void *host, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (bytes == 0) return OFL_OK;
    if (!dptr || !host) return fail(OFL_E_INVALID, "ofl_upload: NULL pointer");
    OFL_HIP(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, stream_of(stream)));
    return OFL_OK;
}

int ofl_download(void *host, const void *dptr, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (bytes == 0) return OFL_OK;
    if (!dptr || !host) return fail(OFL_E_INVALID, "ofl_download: NULL pointer");
    hipStream_t s = stream_of(stream);
    OFL_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    return OFL_OK;
}

int ofl_copy_dev(void *dst, const void *src, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (bytes == 0) return OFL_OK;
    OFL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream_of(stream)));
    return OFL_OK;
}

int ofl_stream_create(void **stream)
{
    OFL_TRY(need_device());
    if (!stream) return fail(OFL_E_INVALID, "ofl_stream_create: NULL");
    hipStream_t s;
    OFL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return OFL_OK;
}

int ofl_stream_destroy(void *stream)
{
    if (!stream) return OFL_OK;
    OFL_TRY(need_device());
    OFL_HIP(hipStreamDestroy((hipStream_t)stream));
    return OFL_OK;
}

int ofl_stream_sync(void *stream)
{
    OFL_TRY(need_device());
    OFL_HIP(hipStreamSynchronize(stream_of(stream)));
    return OFL_OK;
}

int ofl_device_sync(void)
{
    OFL_TRY(need_device());
    OFL_HIP(hipDeviceSynchronize());
    return OFL_OK;
}

int ofl_event_create(void **event)
{
    OFL_TRY(need_device());
    if (!event) return fail(OFL_E_INVALID, "ofl_event_create: NULL");
    hipEvent_t ev;
    OFL_HIP(hipEventCreate(&ev));
    *event = (void *)ev;
    return OFL_OK;
}

int ofl_event_destroy(void *event)
{
    if (!event) return OFL_OK;
    OFL_TRY(need_device());
    OFL_HIP(hipEventDestroy((hipEvent_t)event));
    return OFL_OK;
}

int ofl_event_record(void *event, void *stream)
{
    OFL_TRY(need_device());
    OFL_HIP(hipEventRecord((hipEvent_t)event, stream_of(stream)));
    return OFL_OK;
}

int ofl_event_sync(void *event)
{
    OFL_TRY(need_device());
    OFL_HIP(hipEventSynchronize((hipEvent_t)event));
    return OFL_OK;
}

int ofl_event_elapsed_ms(void *start, void *stop, float *ms)
{
    OFL_TRY(need_device());
    if (!ms) return fail(OFL_E_INVALID, "ofl_event_elapsed_ms: NULL");
    OFL_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return OFL_OK;
}

// PNG row un-filtering (filter types 0-4 of the PNG specification) for the dataset loaders (KITTI flow PNGs, Sintel
// invalid-pixel masks; reference utils.py:426-490 reads them with cv2.imread).  Host code: file decoding, no device.
int ofl_png_unfilter(const uint8_t *raw, size_t raw_bytes, int height, int stride, int bpp, uint8_t *out)
{
    if (!raw || !out || height <= 0 || stride <= 0 || bpp <= 0) return fail(OFL_E_INVALID, "ofl_png_unfilter: bad arguments");
    if (raw_bytes < (size_t)height * ((size_t)stride + 1)) return fail(OFL_E_INVALID, "ofl_png_unfilter: truncated image data");
    for (int y = 0; y < height; ++y) {
        const uint8_t *line = raw + (size_t)y * (stride + 1);
        const int ftype = line[0];
        ++line;
        uint8_t *cur = out + (size_t)y * stride;
        const uint8_t *prev = y ? cur - stride : nullptr;
        if (ftype < 0 || ftype > 4) return fail(OFL_E_INVALID, "ofl_png_unfilter: unknown filter type %d", ftype);
        for (int i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            if (ftype == 1) pred = a;
            else if (ftype == 2) pred = b;
            else if (ftype == 3) pred = (a + b) >> 1;
            else if (ftype == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (uint8_t)((line[i] + pred) & 255);
        }
    }
    return OFL_OK;
}

int ofl_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    OFL_TRY(need_device());
    size_t f = 0, t = 0;
    OFL_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return OFL_OK;
}

}  // extern "C"
