// CU-partitioned streams: the tracker's lane of a multi-GPU step.
// The replicated tracker of an N-GPU step (gom_lstmatcher.py:366-403 run over N x 8 frames on every rank) is a recurrence of
// small DEPENDENT kernels (~18 per long-term match).  On an idle MI355X it costs 0.19 ms per frame; beside the detector it
// cost 0.7 ms, because every one of those launches waits for the detector's resident workgroups (20-120 us each, never
// pre-empted) to free a slot -- stream priority orders dispatch, it does not evict.  A hardware queue with a CU mask gets its
// own compute units: the tracker stream owns a few CUs of every XCD, the detector stream the rest, and neither waits for the
// other's workgroups.
#include "common.h"

/* cu_mask: `words` x 32 bits, bit i = CU i in the driver's enumeration (on MI300-class parts consecutive bits go round the
 * XCDs first, then the shader engines, then the CUs of a shader array: the first 32 bits are one CU per (XCD, shader engine)).
 * The stream is created non-blocking with respect to the NULL stream like every torch stream. */
extern "C" int gom_stream_create_cu_mask(const unsigned* cu_mask, int words, void** stream_out) {
    GOM_CHECK_ARG(cu_mask && words > 0 && stream_out);
    bool any = false;
    for (int i = 0; i < words; ++i) any |= cu_mask[i] != 0;
    GOM_CHECK_ARG(any);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    *stream_out = (void*)s;
    return GOM_OK;
}

extern "C" int gom_stream_destroy(void* stream) {
    if (!stream) return GOM_OK;
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? GOM_OK : GOM_ERR_HIP_BASE + (int)e;
}
