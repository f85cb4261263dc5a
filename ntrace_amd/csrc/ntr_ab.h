/* ntr_ab.h -- entry points that exist only in the A/B build of the library (libntrace_amd_ab.so, `make -C ntrace_amd/csrc ab`,
 * -DNTR_AB): experiments measured against the product on one GPU box.  Nothing here is part of the drop-in boundary. */
#pragma once
#include "ntrace_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tail hand-off counters of the stream's most recent trace launch that ran as wave-private ray pools (closest-hit
 * launches of incoherent batches): counts[0] = ray continuations appended to the queue by waves that
 * left, counts[1] = continuations taken up by other waves (equal once the launch has completed: every ray handed
 * off is finished by another wave -- its visiting order, and so its hit record, is untouched), counts[2] = queue
 * capacity in continuations.  All zero when no such launch ran on `stream`.  Waits for `stream`.  Diagnostic. */
NTR_API int ntr_trace_handoff_counts(void* stream, uint32_t counts[3]);

#ifdef __cplusplus
}
#endif
