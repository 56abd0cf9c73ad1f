// pvs_gate.hip — the reader / writer gate of an index: mutation beside searches is safe inside the library.
//
// The reference mutates while it serves: one writer actor (db/index_writer.rs; embeddings written at db/extraction_write.rs:574-616,
// quant codes upserted at db/vector_quants.rs:1347-1438, rows deleted by ON DELETE CASCADE) beside up to 16 read connections
// (db/connection.rs:235,320-357), and SQLite's snapshot isolation makes every statement see the tables before or after a
// transaction, never in between.  libpvs holds ONE copy of the rows in HBM, and a removal compacts it in place: a scan streaming
// the rows a k_rows_move is rewriting would return a mix.  Until round 6 the header declared the mutating entry points
// "exclusive" and left the exclusion to the host; now it is here:
//
//   * every entry point that reads rows, ids, groups or keys holds the gate SHARED for the call (GateShared at its top; nested
//     calls on the same index by the same thread pass through).  The stream-ordered ones (pvs_search_device,
//     pvs_search_sharded_async) hold it for the enqueue; afterwards the `pending` flag of their context stands for the search;
//   * every entry point that changes them (pvs_index_add*, _remove_rows, _replace_rows*, _set_order_keys, _set_streams) holds it
//     EXCLUSIVELY: it announces itself (new readers wait: writers are not starved by a pool of 16 searching threads), waits
//     until no shared holder is left, then COMPLETES the stream-ordered searches still in flight on their owners' behalf — the
//     device work, the fallbacks, everything pvs_wait would do — and parks the result in the context: the owner's pvs_wait
//     returns it.  (Until round 5 pvs_index_remove_rows called pvs_sync, which consumed other threads' tickets.)
//
// What a concurrent search observes: the index before the mutation or after it.  A stream-ordered search enqueued before a
// mutation is answered over the rows as they were when it was enqueued.  Cost on the search path: two uncontended lock / unlock
// pairs per call.  Sharded searches in flight (a communicator in the picture) are completed by the writer too, which may issue
// the redo exchange of pvs_wait: with several ranks, mutate collectively (every rank at the same point of its program).
#include <thread>

#include "pvs_index.hpp"

namespace {
struct Held {
    const pvs_index *ix;
    int shared, excl;
};
thread_local std::vector<Held> t_held;  // the gates this thread holds (re-entrancy: a nested entry point must not wait for a writer that waits for us)

Held *held_find(const pvs_index *ix) {
    for (Held &h : t_held)
        if (h.ix == ix) return &h;
    return nullptr;
}
void held_drop(Held *h) {
    if (h->shared == 0 && h->excl == 0) {
        *h = t_held.back();
        t_held.pop_back();
    }
}
}  // namespace

void pvs_gate_shared_enter(pvs_index *ix) {
    if (Held *h = held_find(ix)) {  // nested (or inside this thread's own mutation): already covered
        h->shared++;
        return;
    }
    {
        std::unique_lock<std::mutex> lk(ix->mu);
        while (ix->gate_writer_active || ix->gate_writers_waiting) ix->ctx_cv.wait(lk);
        ix->gate_shared++;
    }
    t_held.push_back({ix, 1, 0});
}

void pvs_gate_shared_exit(pvs_index *ix) {
    Held *h = held_find(ix);
    if (!h) return;
    const bool counted = h->excl == 0 && h->shared == 1;  // the outermost shared hold of a thread that is not the writer
    h->shared--;
    held_drop(h);
    if (!counted) return;
    bool wake;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->gate_shared--;
        // (a writer that has announced itself is `waiting` until its turn among writers and `active` while it drains the readers:
        //  it sleeps on ctx_cv in both states — ctx_done's notify comes BEFORE this decrement, so the last reader out must wake it)
        wake = ix->gate_shared == 0 && (ix->gate_writers_waiting || ix->gate_writer_active);
    }
    if (wake) ix->ctx_cv.notify_all();
}

bool pvs_gate_excl_enter(pvs_index *ix) {
    if (Held *h = held_find(ix)) {
        if (h->excl) {  // nested mutation by the writer itself
            h->excl++;
            return true;
        }
        // a thread that holds the gate shared and asks for it exclusively would wait for itself (no entry point of the library does
        // that; a host callback running inside a search might): refused, the caller returns PVS_ERR_STATE
        return false;
    }
    const bool multi = is_multi(ix);
    std::unique_lock<std::mutex> lk(ix->mu);
    ix->gate_writers_waiting++;
    while (ix->gate_writer_active) ix->ctx_cv.wait(lk);  // writers take turns
    ix->gate_writer_active = true;
    ix->gate_writers_waiting--;
    for (;;) {
        // a stream-ordered search in flight nobody is completing: do it here, park the result for its owner
        int found = -1;
        bool busy_elsewhere = false;
        for (uint32_t i = 0; i < NCTX; i++) {
            const bool busy = multi ? ix->mctx[i].busy : ix->ctx[i].busy, pending = multi ? ix->mctx[i].pending : ix->ctx[i].pending;
            const bool draining = multi ? ix->mctx[i].draining : ix->ctx[i].draining;
            if (busy && draining) busy_elsewhere = true;  // its owner is inside pvs_wait: wait for that
            if (busy && pending && !draining && found < 0) found = (int)i;
        }
        if (found >= 0) {
            if (multi)
                ix->mctx[found].draining = true;
            else
                ix->ctx[found].draining = true;
            lk.unlock();
            const pvs_status st = multi ? multi_ticket_complete_(ix, (uint32_t)found) : pvs_ticket_complete_(ix, (uint32_t)found);
            const std::string err = st != PVS_OK ? pvs_last_error() : "";
            lk.lock();
            if (multi) {
                MultiCtx &m = ix->mctx[found];
                m.draining = false, m.pending = false, m.finished = true, m.fin_status = st, m.fin_err = err;
            } else {
                SearchCtx &c = ix->ctx[found];
                c.draining = false, c.pending = false, c.finished = true, c.fin_status = st, c.fin_err = err;
            }
            ix->ctx_cv.notify_all();
            continue;
        }
        if (ix->gate_shared == 0 && !busy_elsewhere) break;
        ix->ctx_cv.wait(lk);
    }
    lk.unlock();
    t_held.push_back({ix, 0, 1});
    return true;
}

void pvs_gate_excl_exit(pvs_index *ix) {
    Held *h = held_find(ix);
    if (!h || !h->excl) return;
    h->excl--;
    if (h->excl) return;
    held_drop(h);
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->gate_writer_active = false;
    }
    ix->ctx_cv.notify_all();
}

// (ix->mu held) the row ids of the index's current rows, on the host.  Keyed on ids_epoch: "same number of rows" is not "same
// rows" once rows can leave and arrive (remove x rows + add x rows left a stale id -> row map behind).
pvs_status pvs_host_ids_locked(pvs_index *ix) {
    if (ix->ids_cache_epoch == ix->ids_epoch && ix->h_ids_cache.size() == ix->n) return PVS_OK;
    ix->ids_cache_epoch = UINT64_MAX;
    ix->h_ids_cache.resize(ix->n);
    if (ix->n) {
        if (is_multi(ix)) {
            PVS_TRY(multi_read_ids(ix, 0, ix->n, ix->h_ids_cache.data(), nullptr));
        } else {
            HIP_TRY(hipSetDevice(ix->device));
            HIP_TRY(hipMemcpy(ix->h_ids_cache.data(), ix->d_ids, ix->n * 8, hipMemcpyDeviceToHost));
        }
    }
    ix->ids_cache_epoch = ix->ids_epoch;
    return PVS_OK;
}
