// Step plans: a recorded list of C-ABI calls replayed by ONE call (include/midivae_hip.h, 'STEP PLANS').
//
// What it replaces: the reference enters the device once per minibatch - `autoencoder.fit` -> the Keras train_function compiled
// once (reference vae_training.py:804-809) - while this engine's step is ~70 launches across 7 queues.  Issued one by one through
// ctypes that is 2.5-3.1 ms of CPython per step; replayed from here it is the launches' own cost (~5 us each).
//
// A plan is NOT a captured graph: every call goes through the same entry point, on the same stream, in the same order as when it
// was recorded - kernels that wait for each other across queues (time-pipelined stacks) need exactly that (a
// replayed hipGraph measured 13.1 ms against 9.5 ms eager in round 1).  Host code only; nothing here touches device memory.
#include "../../include/midivae_hip.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

// ---- one thunk per entry point: arguments decoded from 64-bit slots ----------------------------------------------------------
template <class T>
inline T from_slot(uint64_t v) {
    if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(static_cast<uintptr_t>(v));
    else if constexpr (std::is_same_v<T, float>) {
        const uint32_t bits = static_cast<uint32_t>(v);
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
    } else return static_cast<T>(v);
}
template <class... A, size_t... I>
inline int call_impl(int (*fn)(A...), const uint64_t* s, std::index_sequence<I...>) { return fn(from_slot<A>(s[I])...); }
template <class... A>
inline int call_with(int (*fn)(A...), const uint64_t* s) { return call_impl(fn, s, std::index_sequence_for<A...>{}); }
template <class... A>
constexpr int arity_of(int (*)(A...)) { return static_cast<int>(sizeof...(A)); }

struct entry {
    const char* name;
    int (*thunk)(const uint64_t*);
    int nargs;
};
#define MVAE_ENTRY(f) {#f, [](const uint64_t* s) -> int { return call_with(&f, s); }, arity_of(&f)}
// every stream-taking entry point of the header (the host packers, the queries and the plan functions themselves are not calls
// of a step)
const entry k_entries[] = {
    MVAE_ENTRY(mvae_rnn_fwd), MVAE_ENTRY(mvae_rnn_bwd), MVAE_ENTRY(mvae_rnn_fwd_multi), MVAE_ENTRY(mvae_rnn_bwd_multi),
    MVAE_ENTRY(mvae_pack_recurrent), MVAE_ENTRY(mvae_gemm), MVAE_ENTRY(mvae_gemm_kstream_multi), MVAE_ENTRY(mvae_gemm_multi), MVAE_ENTRY(mvae_colsum),
    MVAE_ENTRY(mvae_stream_wait_value32), MVAE_ENTRY(mvae_stream_write_value32), MVAE_ENTRY(mvae_prepare_batch),
    MVAE_ENTRY(mvae_outer_bias_tile16), MVAE_ENTRY(mvae_gather2_tile16), MVAE_ENTRY(mvae_colsum_weighted),
    MVAE_ENTRY(mvae_sum_over_time), MVAE_ENTRY(mvae_head), MVAE_ENTRY(mvae_latent_fwd), MVAE_ENTRY(mvae_latent_bwd),
    MVAE_ENTRY(mvae_latent_chain_fwd), MVAE_ENTRY(mvae_latent_chain_bwd), MVAE_ENTRY(mvae_relayout), MVAE_ENTRY(mvae_tanh_bwd),
    MVAE_ENTRY(mvae_convert), MVAE_ENTRY(mvae_make_table), MVAE_ENTRY(mvae_transpose_convert), MVAE_ENTRY(mvae_adam_step),
    MVAE_ENTRY(mvae_adam_step_dev), MVAE_ENTRY(mvae_rmsprop_step), MVAE_ENTRY(mvae_scalars_accumulate), MVAE_ENTRY(mvae_copy2d_f32),
    MVAE_ENTRY(mvae_history_from_latent), MVAE_ENTRY(mvae_signature_head_fwd), MVAE_ENTRY(mvae_signature_head_bwd),
    MVAE_ENTRY(mvae_softmax_bwd_add), MVAE_ENTRY(mvae_bi_concat), MVAE_ENTRY(mvae_add_time_reversed),
    MVAE_ENTRY(mvae_event_record), MVAE_ENTRY(mvae_stream_wait_event),
};
constexpr int k_max_slots = 16;

struct blob {
    int slot;
    std::vector<uint64_t> words;      // 8-byte aligned copy of the caller's struct(s)
    size_t bytes;
};
struct call {
    const entry* e;
    uint64_t slots[k_max_slots];
    std::vector<blob> blobs;
};
struct patch {
    int call, slot;
    int64_t offset;      // byte offset into the blob of (call, slot); < 0: the scalar slot itself
    int key;
    int64_t add;
};

}  // namespace

struct mvae_plan {
    std::vector<call> calls;
    std::vector<patch> patches;
    int failed = -1;
};

extern "C" int mvae_event_create(void** event) {
    if (!event) return MVAE_E_ARG;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MVAE_E_LAUNCH;
    *event = e;
    return MVAE_OK;
}
extern "C" int mvae_event_create_timed(void** event) {
    if (!event) return MVAE_E_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return MVAE_E_LAUNCH;
    *event = e;
    return MVAE_OK;
}
extern "C" int mvae_event_elapsed_ms(void* first, void* second, float* ms) {
    if (!first || !second || !ms) return MVAE_E_ARG;
    if (hipEventSynchronize(reinterpret_cast<hipEvent_t>(second)) != hipSuccess) return MVAE_E_LAUNCH;
    return hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(first), reinterpret_cast<hipEvent_t>(second)) == hipSuccess
               ? MVAE_OK
               : MVAE_E_LAUNCH;
}
extern "C" int mvae_event_destroy(void* event) {
    if (!event) return MVAE_E_ARG;
    return hipEventDestroy(reinterpret_cast<hipEvent_t>(event)) == hipSuccess ? MVAE_OK : MVAE_E_LAUNCH;
}
extern "C" int mvae_event_record(void* event, void* stream) {
    if (!event) return MVAE_E_ARG;
    return hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? MVAE_OK
                                                                                                                  : MVAE_E_LAUNCH;
}
extern "C" int mvae_event_synchronize(void* event) {
    if (!event) return MVAE_E_ARG;
    return hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)) == hipSuccess ? MVAE_OK : MVAE_E_LAUNCH;
}
extern "C" int mvae_stream_wait_event(void* stream, void* event) {
    if (!event) return MVAE_E_ARG;
    return hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0) == hipSuccess
               ? MVAE_OK
               : MVAE_E_LAUNCH;
}

extern "C" int mvae_plan_create(mvae_plan** out) {
    if (!out) return MVAE_E_ARG;
    *out = new (std::nothrow) mvae_plan();
    return *out ? MVAE_OK : MVAE_E_LAUNCH;
}
extern "C" int mvae_plan_destroy(mvae_plan* p) {
    if (!p) return MVAE_E_ARG;
    delete p;
    return MVAE_OK;
}
extern "C" int mvae_plan_add_call(mvae_plan* p, const char* entry_point, const uint64_t* slots, int32_t n_slots) {
    if (!p || !entry_point || n_slots < 0 || n_slots > k_max_slots || (n_slots && !slots)) return MVAE_E_ARG;
    for (const entry& e : k_entries) {
        if (std::strcmp(e.name, entry_point) != 0) continue;
        if (e.nargs != n_slots) return MVAE_E_ARG;
        call c;
        c.e = &e;
        std::memset(c.slots, 0, sizeof(c.slots));
        for (int i = 0; i < n_slots; ++i) c.slots[i] = slots[i];
        p->calls.push_back(std::move(c));
        return static_cast<int>(p->calls.size()) - 1;
    }
    return MVAE_E_ARG;
}
extern "C" int mvae_plan_set_blob(mvae_plan* p, int32_t ci, int32_t slot, const void* data, size_t bytes) {
    if (!p || ci < 0 || ci >= static_cast<int>(p->calls.size()) || !data || bytes == 0) return MVAE_E_ARG;
    call& c = p->calls[ci];
    if (slot < 0 || slot >= c.e->nargs) return MVAE_E_ARG;
    for (const blob& b : c.blobs)
        if (b.slot == slot) return MVAE_E_ARG;
    blob b;
    b.slot = slot;
    b.bytes = bytes;
    b.words.assign((bytes + 7) / 8, 0);
    std::memcpy(b.words.data(), data, bytes);
    c.blobs.push_back(std::move(b));
    return MVAE_OK;
}
extern "C" int mvae_plan_add_patch(mvae_plan* p, int32_t ci, int32_t slot, int64_t offset, int32_t key, int64_t add) {
    if (!p || ci < 0 || ci >= static_cast<int>(p->calls.size()) || key < 0) return MVAE_E_ARG;
    const call& c = p->calls[ci];
    if (slot < 0 || slot >= c.e->nargs) return MVAE_E_ARG;
    if (offset >= 0) {
        bool ok = false;
        for (const blob& b : c.blobs)
            if (b.slot == slot && offset % 4 == 0 && static_cast<size_t>(offset) + 4 <= b.bytes) ok = true;
        if (!ok) return MVAE_E_ARG;
    }
    p->patches.push_back(patch{ci, slot, offset, key, add});
    return MVAE_OK;
}
extern "C" int mvae_plan_size(const mvae_plan* p) { return p ? static_cast<int>(p->calls.size()) : MVAE_E_ARG; }
extern "C" int mvae_plan_failed_call(const mvae_plan* p) { return p ? p->failed : MVAE_E_ARG; }

extern "C" int mvae_plan_run(mvae_plan* p, int32_t first, int32_t last, const uint64_t* key_values, int32_t n_keys) {
    if (!p) return MVAE_E_ARG;
    const int n = static_cast<int>(p->calls.size());
    if (last < 0) last = n;
    if (first < 0 || first > last || last > n) return MVAE_E_ARG;
    for (const patch& pt : p->patches) {
        if (pt.call < first || pt.call >= last) continue;
        if (pt.key >= n_keys || !key_values) return MVAE_E_ARG;
        const uint32_t v = static_cast<uint32_t>(key_values[pt.key] + static_cast<uint64_t>(pt.add));
        call& c = p->calls[pt.call];
        if (pt.offset < 0) c.slots[pt.slot] = v;
        else
            for (blob& b : c.blobs)
                if (b.slot == pt.slot) std::memcpy(reinterpret_cast<unsigned char*>(b.words.data()) + pt.offset, &v, 4);
    }
    p->failed = -1;
    for (int i = first; i < last; ++i) {
        call& c = p->calls[i];
        for (blob& b : c.blobs) c.slots[b.slot] = reinterpret_cast<uint64_t>(b.words.data());
        const int rc = c.e->thunk(c.slots);
        if (rc != 0) {
            p->failed = i;
            return rc;
        }
    }
    return MVAE_OK;
}
