/* luminair_hip_batch.h - the additional entry points of libluminair_hip_batch.so (the lock-step batch build of the same
 * sources, luminair_amd/csrc/batch.h).  That library also exports the whole ABI of luminair_hip.h. */
#ifndef LUMINAIR_HIP_BATCH_H
#define LUMINAIR_HIP_BATCH_H

#include "luminair_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Lock-step batches of SMALL proofs.
 * Proofs whose tables have a few thousand rows - BASELINE config 4 (examples/black-schole-nn/src/main.rs:61-103), the
 * reference's own benchmark shape (crates/graph/benches/ops.rs:92-166) - are bound by their ~60 kernel launches and 6
 * host round trips.  lmn_batch_prove proves n <= slots pies of IDENTICAL SHAPE (same table kinds, same row counts) with
 * one launch per pipeline step and one host wait per transcript step for the whole batch; every proof is byte-identical
 * to what lmn_prove returns for its pie.  tables[i] = the n_tables tables of pie i; proofs[i] / lens[i] receive
 * lmn_free-able bytes; rcs (may be NULL) the per-pie status.  Returns LMN_OK or the first failing pie's code; pies of
 * different shapes -> LMN_ERR_INVALID_ARGUMENT.
 * Threads: calls on ONE batch object are serialised by the object; DIFFERENT batch objects are independent and may be
 * driven from different threads at the same time - three objects of 64 slots make 30 k instead of 21 k proofs/s on the
 * reference's benchmark shape (three of 192 slots: 46 - 49 k; keep the worker threads of all batch objects of a process - LMN_BATCH_THREADS each, default 8 - at 24 or fewer), because one group's host code overlaps another group's launches. */
typedef struct lmn_batch lmn_batch;
int lmn_batch_create(int device, const lmn_config* cfg, uint32_t slots, lmn_batch** out);
int lmn_batch_prove(lmn_batch* batch, uint32_t n, const lmn_table* const* tables, size_t n_tables,
                    const lmn_settings* settings, uint8_t** proofs, size_t* lens, int* rcs);
const char* lmn_batch_last_error(const lmn_batch* batch);
uint64_t lmn_batch_counter(const lmn_batch* batch, int which); /* so far - 0: batched kernel launches, 1: host waits, 2: batched transfer launches, 3: transfers issued one by one */
void lmn_batch_destroy(lmn_batch* batch);

#ifdef __cplusplus
}
#endif
#endif
