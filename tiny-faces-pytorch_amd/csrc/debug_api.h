/* debug_api.h -- debugging and measurement probes of libtinyfaces_hip.so.  NOT part of the C ABI (include/tinyfaces_hip.h): exported for
 * the test-suite (tests/test_gpu_small_ops.py) and the measurement scripts (scripts/contention.py, floor.py, trace_conv3x3h.py) only. */
#ifndef TINYFACES_HIP_DEBUG_API_H
#define TINYFACES_HIP_DEBUG_API_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* debugging hook of the halo-resident 3x3 kernel (csrc/conv3x3h.hip): register (NULL: clear) a DEVICE buffer of
 * 8 blocks x 8 waves x 64 stages x 8 uint64; the next launches run an instrumented instantiation that stamps s_memtime at the
 * five points of every K stage (scripts/trace_conv3x3h.py).  Not part of the product path. */
int tf_debug_conv3x3h_trace(void* device_buf);
/* rows of the output tile (4 or 6) the most recent conv3x3h launch of this process used (r6: tests/test_gpu_conv.py) */
int tf_debug_conv3x3h_tile_rows(void);
/* interference probe of the two-stream contention measurement (csrc/probe.hip, scripts/contention.py): `blocks` workgroups of 256
 * threads that hog ONE CU resource for `iters` rounds -- kind 0 park (LDS capacity + wave slots only), 1 L2 loads, 2 HBM loads,
 * 3 MFMA, 4 LDS-DMA, 5 fp32 atomics, 6 LDS reads, 9 (r5) operand-streaming pattern probe (iters = run bytes | depth << 16), 10 (r5) tile-shaped epilogue traffic alone (iters = input matrices), 7 (r4) `iters` device-wide barriers in one launch (blocks <= 1024); `buf` / `window_bytes`: device window of the memory kinds.  Not part of the
 * product path. */
int tf_debug_probe(int kind, int blocks, int lds_bytes, void* buf, size_t window_bytes, int iters, void* stream);
int tf_debug_probe_chain(int kind, int blocks, int lds_bytes, void* buf, size_t window_bytes, int iters, int repeat, void* stream);   /* `repeat` launches from one host call */
/* test hook: raw lane mapping of ds_read_b64_tr_b16 (see tests/test_gpu_small_ops.py) */
int tf_probe_tr16(unsigned short* out256, void* stream);

#ifdef __cplusplus
}
#endif
#endif
