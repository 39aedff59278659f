// vf_table_dev.h -- the device count table: slot protocol, find-or-insert, the workgroup aggregator, the count log's writer side and
// the workgroup key cache (used by k_finish, k_generic, k_merge, k_rehash, k_export: vf_k_table.hip)
#pragma once
#include "vf_kernels_common.h"

namespace vf {

// ------------------------------------------------------------------------------------------------
// device count table
// ------------------------------------------------------------------------------------------------
struct Hash { uint32_t h1, h2; };
__device__ __forceinline__ void hash_init(Hash &h, uint32_t meta) {
    h.h1 = 0x811C9DC5u ^ meta;
    h.h2 = 0x9E3779B9u + meta * 0x85EBCA6Bu;
}
__device__ __forceinline__ void hash_step(Hash &h, uint32_t k) {
    h.h1 = (h.h1 ^ k) * 0x01000193u;
    h.h2 = (h.h2 + k) * 0x9E3779B1u;
    h.h2 ^= h.h2 >> 15;
}
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void hash_fin(Hash &h) { h.h1 = fmix32(h.h1 ^ (h.h2 >> 7)); h.h2 = fmix32(h.h2) | 1u; }

__device__ __forceinline__ uint32_t key_dwords(uint32_t kind, uint32_t len) {
    return kind == kKindPacked ? (len + 15u) >> 4 : (len + 3u) >> 2;
}

constexpr int kInlineDwords = 14;        // key dwords held inside the slot

// key held in registers (static indexing)
template <int ND> struct RegKey {
    static constexpr uint32_t kMaxDwords = (uint32_t)ND;      // (a key held in ND registers cannot be longer: the LONG-key paths compile out)
    const uint32_t (&k)[ND];
    template <class F> __device__ __forceinline__ void each(uint32_t nd, F f) const {
#pragma unroll
        for (int d = 0; d < ND; ++d) if ((uint32_t)d < nd) f((uint32_t)d, k[d]);
    }
    // first kInlineDwords dwords, zero padded (register keys are zero beyond their length)
    __device__ __forceinline__ void head(uint32_t, uint32_t (&o)[kInlineDwords]) const {
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) o[d] = d < ND ? k[d < ND ? d : 0] : 0u;
    }
};
// key held in memory, element d at p[d*stride]
struct MemKey {
    static constexpr uint32_t kMaxDwords = 0xFFFFFFFFu;
    const uint32_t *p;
    uint64_t stride;
    uint32_t mask = 0xFFFFFFFFu;             // (a key read out of a tagged slot line: ~kTagBits)
    template <class F> __device__ __forceinline__ void each(uint32_t nd, F f) const {
        for (uint32_t d = 0; d < nd; ++d) f(d, p[d * stride] & mask);
    }
    __device__ __forceinline__ void head(uint32_t nd, uint32_t (&o)[kInlineDwords]) const {
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) o[d] = (uint32_t)d < nd ? (p[d * stride] & mask) : 0u;
    }
};

// ---- device count table: one 64-byte slot per key ---------------------------------------------------
//   slot = { u64 w0; u32 key[14] }                       (a lookup touches exactly one 64-byte line)
//   w0   = tag | len<<32 | kind<<48 | LONG<<50 | OVF<<51 | READY<<63 ;  0 = empty
// Keys of up to 14 dwords (56 peptide letters / 224 bases) live in the slot, zero padded, so a probe
// compares all 14 dwords without predication.  Longer keys (LONG) keep their arena offset in key[0..1].
// Counts sit in their own array: atomics execute at the memory side and drop the line from L2, so a
// count sharing the slot's line made every probe of a hot key miss L2 (measured 3x slower).
// Protocol: claim by 64-bit CAS on w0 (READY clear), write the key, s_waitcnt vmcnt(0), store w0|READY.
// Every word another workgroup may read is written and read with agent-scope (sc1) accesses; all of
// them are write-once, so a stale view can only ever cause a duplicate entry, never a false match
// (the host sums duplicate strings when it builds the result).  A negative compare is only believed
// when it comes from a load issued after READY had already been observed.
constexpr uint64_t kW0Ready = 1ull << 63, kW0Long = 1ull << 50, kW0Ovf = 1ull << 51;
// kW0One: the entry's count is one more than its count words say - the read that inserted the key left its own count in
// the claim instead of a count word (an insert then costs no count line: in a cleared table that is one random line read
// and written back less for every new key, 1.6 M of the 8.7 M random lines of a cfg-2 step).  Set only by the claim, so
// nothing ever has to change it; every reader of counts adds it (count_of).
constexpr uint64_t kW0One = 1ull << 52;
// Epoch tags (Table::epoch, vf_device.h): w0 bits 53..56.  A slot whose w0 is 0 or carries another epoch is FREE (claimed by
// a CAS that expects exactly the stale word); a letter key's dwords carry the epoch in their bytes' top bits, so a dword left
// over from an earlier epoch - or never written - reads as "the owner's store has not landed yet", exactly what a zero
// dword means in an untagged table.  Untagged tables (epoch 0: packed-DNA keys have no spare bits) are cleared by zeroing.
constexpr int kW0EpochShift = 53;
constexpr uint64_t kW0EpochMask = 15ull << kW0EpochShift;
__device__ __forceinline__ uint64_t w0_epoch(const Table &t) { return (uint64_t)t.epoch << kW0EpochShift; }
__device__ __forceinline__ bool w0_free(const Table &t, uint64_t w0) { return w0 == 0ull || (w0 & kW0EpochMask) != w0_epoch(t); }
__device__ __forceinline__ bool w0_live(const Table &t, uint64_t w0) { return (w0 & kW0Ready) && !(w0 & kW0Ovf) && (w0 & kW0EpochMask) == w0_epoch(t); }
// has the owner's store of this key dword landed?
__device__ __forceinline__ bool dword_written(const Table &t, uint32_t v) { return t.key_tag ? (v & kTagBits) == t.key_tag : v != 0u; }
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
// A LONG line's key[0..1]: the key's arena offset (36 bits: the arena ends at 2^36 dwords) and, above it, 28 check bits
// made of the line's tag and the table's epoch - never 0.  Insurance: nothing read from a slot line is used as an address
// unless it validates itself and lies inside the arena.  (Round 3's fault was a reader that paired the READY w0 a lost CAS
// had returned with the key words of the load BEFORE the CAS - table_find_or_insert, image_older - and followed the letters
// an earlier epoch had left in key[0..1]; while tables were cleared by zeroing the stale word was 0, the compare against
// the key at arena[0] failed and the reader inserted a duplicate, never noticed.  The hardware keeps its side: a writer's
// store, s_waitcnt, flag store are seen in that order from another XCD - scripts/ubench/visibility.hip, 1.7 x 10^8 watched
// transitions per variant, none torn.)  A reader that finds the word wanting looks again.
constexpr uint64_t kLongOffMask = (1ull << 36) - 1ull;
__device__ __forceinline__ uint32_t long_check(const Table &t, uint32_t tag) { return (((tag >> 4) ^ t.epoch) & 0x07FFFFFFu) | 0x08000000u; }
__device__ __forceinline__ uint64_t long_word(const Table &t, uint32_t tag, uint64_t off) { return off | ((uint64_t)long_check(t, tag) << 36); }
__device__ __forceinline__ bool long_word_ok(const Table &t, uint32_t tag, uint64_t word, uint32_t nd) {
    return (uint32_t)(word >> 36) == long_check(t, tag) && (word & kLongOffMask) + nd <= t.arena_cap;
}
// A volatile access to LDS through a GENERIC pointer compiles to a FLAT instruction (the compiler gives up on the address
// space), and FLAT accesses to LDS travel through the vector-memory path: they are not ordered with the DS instructions of
// the same wave.  k_finish read a line stage it had just written with ds_write through flat_load (27 FLAT instructions in
// the kernel until round 3) - never seen to go wrong, but a stale read there is a wild global store.  Every volatile LDS
// access goes through an explicit LDS pointer: DS instructions, executed in issue order.
#define VF_LDS(T, p) ((volatile __attribute__((address_space(3))) T *)(p))
#ifdef VF_BOUNDS_CHECK      // diagnosis build only (make variant NAME=bc DEFS=-DVF_BOUNDS_CHECK): the first violated bound is
// recorded in the debug counters (code, two values, workgroup) and the access is skipped
__device__ __forceinline__ bool vf_bound(const Table &t, bool ok, uint64_t code, uint64_t v0, uint64_t v1) {
    if (!ok && atomicCAS((unsigned long long *)&t.counters[C_DBG0], 0ull, (unsigned long long)code) == 0ull) {
        t.counters[C_DBG1] = v0; t.counters[C_DBG2] = v1; t.counters[C_DBG3] = blockIdx.x;
    }
    return ok;
}
#define VF_BOUND(t, ok, code, v0, v1) vf_bound(t, ok, code, v0, v1)
#else
#define VF_BOUND(t, ok, code, v0, v1) true
#endif

__device__ __forceinline__ void slot_load(const uint4 *p, uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
// whole-slot write-through store: four 16-byte sc1 stores (8-byte sc1 stores are one fabric write each
// and cost 2.7x per byte, MI355X_MICROARCH "stores of each flavour")
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void slot_store(uint4 *p, u32x4 a, u32x4 b, u32x4 c, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:32 sc1\n\t"
                 "global_store_dwordx4 %0, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(p), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
__device__ __forceinline__ void slot_store_nowait(uint4 *p, u32x4 a, u32x4 b, u32x4 c, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:32 sc1\n\t"
                 "global_store_dwordx4 %0, %4, off offset:48 sc1"
                 :: "v"(p), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
// Counts: one 64-bit word per slot, apart from the slot lines, and k_finish never touches it.  An occurrence of a key
// that is already in the table is written to the batch's count log as the key's slot index (CountLog, vf_device.h) and
// applied after the kernel, range by range (k_log_apply).  Rounds 1-2 counted in place, with an atomic per occurrence
// executed in the issuing XCD's L2 on one count array per XCD: every tail key then cost a count line fetched and written
// back (384 MB of the 975 MB k_finish moved per 10 M reads) and - loads, stores and atomics of a wave retire in issue
// order - every trip's slot lookup waited behind the count atomics of the trip before it.
// A count of any size from anywhere (merge, growth, the byte-wise path): device scope, executed at the memory side.
__device__ __forceinline__ void base_add(const Table &t, uint32_t slot, unsigned long long v) {
    __hip_atomic_fetch_add(t.counts + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t count_of(const Table &t, uint64_t slot, uint64_t w0) {      // the entry's count (see kW0One)
    return t.counts[slot] + ((w0 & kW0One) ? 1ull : 0ull);
}

// ---- workgroup-local state: arena chunks for LONG keys, insert counters --------------------------
// LONG keys take arena space from a chunk owned by the workgroup; only a refill touches the global
// bump pointer.  The chunk state survives across launches in t.chunk_state[row][block], one row per
// stream, so the arena loses at most one partial chunk per (stream, workgroup) in total.
//   state = (chunk_index + 1) << 32 | dwords_used      (0 = no chunk yet)
struct BlockTable {
    unsigned long long chunk;                // LDS copy of the chunk state
    uint32_t new_entries;                    // keys inserted by this workgroup (flushed at kernel end)
    uint32_t new_key_dwords;                 // their total key dwords
};
__device__ __forceinline__ void block_table_begin(BlockTable &bt, const Table &t) {
    if (threadIdx.x == 0) {
        bt.chunk = t.chunk_state[(uint64_t)t.chunk_row * kMaxGrid + blockIdx.x];
        if (!VF_BOUND(t, (bt.chunk >> 32) <= t.arena_cap / kChunk && (uint32_t)bt.chunk <= (1u << 24), 1, bt.chunk, t.chunk_row)) bt.chunk = 0;
        bt.new_entries = 0;
        bt.new_key_dwords = 0;
    }
}
__device__ __forceinline__ void block_table_end(BlockTable &bt, const Table &t) {     // after a __syncthreads()
    if (threadIdx.x == 0) {
        t.chunk_state[(uint64_t)t.chunk_row * kMaxGrid + blockIdx.x] = bt.chunk;
        if (bt.new_entries) {
            atomicAdd((unsigned long long *)&t.counters[C_ENTRIES], (unsigned long long)bt.new_entries);
            atomicAdd((unsigned long long *)&t.counters[C_KEY_DWORDS], (unsigned long long)bt.new_key_dwords);
        }
    }
}
enum { kAllocRetry = 0, kAllocOk = 1, kAllocFull = 2 };
__device__ __forceinline__ int arena_alloc(BlockTable &bt, const Table &t, uint32_t nd, uint64_t &off) {
    if (nd > kChunk / 4) {                   // oversized key: whole chunks straight from the global pointer
        const uint64_t span = (uint64_t)(nd + kChunk - 1) / kChunk * kChunk;
        off = atomicAdd((unsigned long long *)&t.counters[C_ARENA_TOP], (unsigned long long)span);
        return off + span <= t.arena_cap ? kAllocOk : kAllocFull;
    }
    const unsigned long long old = atomicAdd(&bt.chunk, (unsigned long long)nd);
    const uint32_t used = (uint32_t)old, idxp1 = (uint32_t)(old >> 32);
    if (idxp1 != 0u && used + nd <= kChunk) { off = (uint64_t)(idxp1 - 1u) * kChunk + used; return kAllocOk; }
    const bool crosser = idxp1 == 0u ? used == 0u : used <= kChunk;
    if (!crosser) return kAllocRetry;        // the lane that crossed the chunk end is fetching a new chunk
    const uint64_t base = atomicAdd((unsigned long long *)&t.counters[C_ARENA_TOP], (unsigned long long)kChunk);
    if (base + kChunk > t.arena_cap) {
        atomicExch(&bt.chunk, 0ull);         // let a later lane try again (and fail the same way)
        return kAllocFull;
    }
    atomicExch(&bt.chunk, ((unsigned long long)(base / kChunk + 1u) << 32) | (unsigned long long)nd);
    off = base;
    return kAllocOk;
}

__device__ __forceinline__ void slot_load_key(const uint4 *p, uint32_t nd, bool can, uint32_t tag, uint4 &a, uint4 &b, uint4 &c, uint4 &d);
#ifndef VF_SLOW_PARTIAL
#define VF_SLOW_PARTIAL 1       // (A/B: make variant NAME=slowfull DEFS=-DVF_SLOW_PARTIAL=0: whole slot lines in the insert queue, as until round 4)
#endif
// Finds key's slot, inserting the key if absent; returns kNoSlot on a capacity error.
// have_off: arena offset the (LONG) key already occupies (table growth re-inserts entries whose keys stay where they
// are), kNoOff = the key is copied into a fresh arena allocation.
// one: the caller is counting ONE occurrence; if this lane is the one that inserts the key, the claim carries that count
// (kW0One) and `counted` comes back true - the caller then adds nothing.
constexpr uint64_t kNoOff = ~0ull;
template <class K>
__device__ __forceinline__ uint32_t table_find_or_insert(const Table &t, BlockTable &bt, const K &key, uint32_t kind, uint32_t len, bool active,
                                                         uint64_t have_off, bool one, bool &counted) {
    counted = false;
    uint32_t found = kNoSlot;
    const uint32_t nd = key_dwords(kind, len);
    const uint32_t meta = len | (kind << 16);
    Hash h;
    hash_init(h, meta);
    key.each(nd, [&](uint32_t, uint32_t v) { hash_step(h, v); });
    hash_fin(h);
    const bool is_long = K::kMaxDwords > (uint32_t)kInlineDwords && nd > (uint32_t)kInlineDwords;
    const uint64_t w0mine = (uint64_t)h.h2 | ((uint64_t)meta << 32) | (is_long ? kW0Long : 0ull) | w0_epoch(t);
    const uint64_t w0pub = w0mine | (one ? kW0One : 0ull);                // what this lane writes if it inserts
    // Letter keys (kind 0) validate themselves: every dword of a written key is non-zero, so a slot can
    // be claimed READY at once and its key stored without waiting; a reader that meets a zero dword
    // inside the key's length knows the line is still in flight and looks again.  Packed-DNA keys may
    // hold zero dwords and keep the claim -> store -> wait -> READY order.
    const bool selfval = kind == kKindBytes && !is_long;
    const uint64_t w0claim = selfval ? (w0pub | kW0Ready) : w0pub;
    uint32_t kin[kInlineDwords];
    key.head(nd, kin);
    if (selfval) {                           // the slot's image of a letter key carries the epoch in every dword (padding too)
#pragma unroll
        for (int d = 0; d < kInlineDwords; ++d) kin[d] |= t.key_tag;
    }
    uint32_t slot = h.h1 & t.slot_mask;
    uint32_t probes = 0;
    bool done = !active;
    bool claimed = false;                    // this lane won the slot and must publish its key
    bool confirmed = false;                  // READY already seen at this slot (a mismatch is now final)
    // The loop condition is a wave-wide vote: no lane leaves before every lane is done.  A lane that
    // finds a slot claimed but not yet published simply looks again on the next trip; because the
    // claiming lane (possibly in this very wave) publishes inside the loop body, before the vote,
    // the wait cannot deadlock and the compiler cannot sink the publish past the loop exit.
#ifdef VF_INSERT_COUNTERS
    if (lane_id() == 0) atomicAdd((unsigned long long *)&t.counters[C_DBG0], 1ull);
#endif
    while (__any(!done)) {
#ifdef VF_INSERT_COUNTERS
        if (lane_id() == 0) atomicAdd((unsigned long long *)&t.counters[C_DBG1], 1ull);
#endif
        if (!done) {
            uint4 *sp = t.slots + (uint64_t)slot * 4;
            bool advance = false;
            uint4 l0 = make_uint4(0, 0, 0, 0), l1 = l0, l2 = l0, l3 = l0;
            uint64_t w0 = 0;
            // the w0 a lost CAS returns is NEWER than the line image loaded before it: another lane may have claimed the slot,
            // written its key and set READY in between.  Such a w0 may say "another key" (advance) or "not READY yet" (wait), but
            // the key words that came with the earlier load must not be compared or followed under it (round 3: a LONG key's
            // reader took the letters an earlier epoch had left in key[0..1] for an arena offset).
            bool image_older = false;
            if (!claimed) {
                // Only the 16-byte quarters of the line this lane's key reaches (a LONG key: w0 and the offset word), as in the
                // one-shot lookup: a quarter is one request at the L2, and the queue's loop was 4 of them a trip for every lane
                // in it (round 4's request table: ~5 M of k_finish's 18.8 M read requests a step).  A slot whose w0 matches holds
                // a key of the same length, padded like this one, so the quarters left out compare equal by construction.
                if (VF_SLOW_PARTIAL) slot_load_key(sp, is_long ? 2u : nd, true, selfval ? t.key_tag : 0u, l0, l1, l2, l3);
                else slot_load(sp, l0, l1, l2, l3);
                w0 = (uint64_t)l0.x | ((uint64_t)l0.y << 32);
                if (w0_free(t, w0)) {
                    const uint64_t seen = w0;                    // (0, or a word of an earlier epoch: the CAS expects exactly it)
                    w0 = atomicCAS(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)seen, (unsigned long long)w0claim);
                    claimed = w0 == seen;
                    image_older = !claimed;
#ifdef VF_NO_IMAGE_GUARD        // diagnosis build only (with -DVF_LONG_DIAG): round 3's behaviour, the offset check alone catches it
                    image_older = false;
#endif
#ifdef VF_INSERT_COUNTERS
                    atomicAdd((unsigned long long *)&t.counters[C_DBG2], 1ull);
                    if (claimed) atomicAdd((unsigned long long *)&t.counters[C_DBG3], 1ull);
#endif
                }
            }
            if (claimed) {
                unsigned long long *kw = reinterpret_cast<unsigned long long *>(sp) + 1;
                if (selfval) {
                    slot_store_nowait(sp, u32x4{(uint32_t)w0claim, (uint32_t)(w0claim >> 32), kin[0], kin[1]}, u32x4{kin[2], kin[3], kin[4], kin[5]},
                                      u32x4{kin[6], kin[7], kin[8], kin[9]}, u32x4{kin[10], kin[11], kin[12], kin[13]});
                    atomicAdd(&bt.new_entries, 1u);
                    atomicAdd(&bt.new_key_dwords, nd);
                    found = slot;
                    counted = one;
                    done = true;
                } else if (!is_long) {
                    // the line is ours (w0 claimed, READY clear): rewrite it whole, then publish
                    slot_store(sp, u32x4{(uint32_t)w0pub, (uint32_t)(w0pub >> 32), kin[0], kin[1]}, u32x4{kin[2], kin[3], kin[4], kin[5]},
                               u32x4{kin[6], kin[7], kin[8], kin[9]}, u32x4{kin[10], kin[11], kin[12], kin[13]});
                    st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0pub | kW0Ready));
                    atomicAdd(&bt.new_entries, 1u);
                    atomicAdd(&bt.new_key_dwords, nd);
                    found = slot;
                    counted = one;
                    done = true;
                } else {
                    uint64_t off = have_off;
                    const int rc = have_off != kNoOff ? (int)kAllocOk : arena_alloc(bt, t, nd, off);
                    if (rc == kAllocOk && !VF_BOUND(t, off + nd <= t.arena_cap, 2, off, nd)) {
                        done = true;
                    } else if (rc == kAllocOk) {
                        if (have_off == kNoOff) key.each(nd, [&](uint32_t d, uint32_t v) { st_agent(&t.arena[off + d], v); });
                        st_agent(kw, (unsigned long long)long_word(t, h.h2, off));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0pub | kW0Ready));
                        atomicAdd(&bt.new_entries, 1u);
                        atomicAdd(&bt.new_key_dwords, nd);
                        found = slot;
                        counted = one;
                        done = true;
                    } else if (rc == kAllocFull) {
                        atomicOr((unsigned long long *)&t.counters[C_ERROR], 1ull);
                        st_agent(reinterpret_cast<unsigned long long *>(sp), (unsigned long long)(w0mine | kW0Ready | kW0Ovf));
                        done = true;
                    }
                }
            } else if ((w0 & ~(kW0Ready | kW0Ovf | kW0One)) == w0mine) {
                if (!(w0 & kW0Ready) || image_older) {
                    // claimed by another lane, key not yet published - or published after this trip's load: look at the same
                    // slot again next trip
                } else if (w0 & kW0Ovf) {
                    done = true;
                } else {
                    uint32_t diff = 0;
                    bool offset_landed = true;           // (LONG lines: long_word_ok)
                    if (!is_long) {
                        diff = (l0.z ^ kin[0]) | (l0.w ^ kin[1]) | (l1.x ^ kin[2]) | (l1.y ^ kin[3]) | (l1.z ^ kin[4]) | (l1.w ^ kin[5]) |
                               (l2.x ^ kin[6]) | (l2.y ^ kin[7]) | (l2.z ^ kin[8]) | (l2.w ^ kin[9]) | (l3.x ^ kin[10]) | (l3.y ^ kin[11]) |
                               (l3.z ^ kin[12]) | (l3.w ^ kin[13]);
                    } else {
                        const uint64_t word = (uint64_t)l0.z | ((uint64_t)l0.w << 32);
                        const uint64_t off = word & kLongOffMask;
                        offset_landed = long_word_ok(t, h.h2, word, nd);
#ifdef VF_LONG_DIAG             // diagnosis build only: [C_DBG0] offset words that did not validate, [C_DBG1] LONG lines compared
                        atomicAdd((unsigned long long *)&t.counters[offset_landed ? C_DBG1 : C_DBG0], 1ull);
#endif
                        if (offset_landed) key.each(nd, [&](uint32_t d, uint32_t v) { diff |= ld_agent(&t.arena[off + d]) ^ v; });
                    }
                    if (!offset_landed) {
                        // READY seen, the offset word not yet: the same slot again next trip
                    } else if (diff == 0u) { found = slot; done = true; }
                    else if (selfval) {
                        // a zero dword inside the key's length = the owner's stores are still in flight
                        const uint32_t ln[kInlineDwords] = {l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w};
                        bool incomplete = false;
#pragma unroll
                        // (in a tagged table the padding dwords carry the epoch too: a stale one means the line is still landing)
                        for (int d = 0; d < kInlineDwords; ++d) incomplete = incomplete || (((uint32_t)d < nd || t.key_tag != 0u) && !dword_written(t, ln[d]));
                        if (!incomplete) advance = true;
                    }
                    else if (!confirmed && !is_long) confirmed = true;      // re-read once now that READY is known
                    else advance = true;
                }
            } else {
                advance = true;
            }
            if (advance) {
                confirmed = false;
                slot = (slot + 1u) & t.slot_mask;
                if (++probes > t.slot_mask) {
                    atomicOr((unsigned long long *)&t.counters[C_ERROR], 2ull);
                    done = true;
                }
            }
        }
    }
    return found;
}

template <class K>
__device__ __forceinline__ void table_add(const Table &t, BlockTable &bt, const K &key, uint32_t kind, uint32_t len, uint64_t cnt, bool active,
                                          uint64_t have_off = kNoOff) {
    bool counted;
    const bool one = cnt == 1ull;
    const uint32_t slot = table_find_or_insert(t, bt, key, kind, len, active, have_off, one, counted);
    if (slot != kNoSlot && !counted) base_add(t, slot, (unsigned long long)cnt);
}

// Workgroup-level count aggregation: a small LDS table keyed by slot index collects the +1s of a
// whole workgroup (all its grid-stride iterations); what it holds at kernel end goes to the count log as
// (slot, count) pairs, so the Zipf head of the variant distribution costs one log entry per workgroup
// instead of one per read.  A slot that loses the LDS entry to another slot is logged occurrence by occurrence.
#ifndef VF_AGG_SIZE
#define VF_AGG_SIZE 1024
#endif
constexpr int kAggSize = VF_AGG_SIZE;
struct Agg {
    uint32_t slot[kAggSize];
    uint32_t cnt[kAggSize];
};
__device__ __forceinline__ void agg_init(Agg &a) {
    for (int i = threadIdx.x; i < kAggSize; i += kBlock) { a.slot[i] = kNoSlot; a.cnt[i] = 0; }
}
// true = counted here
__device__ __forceinline__ bool agg_try(Agg &a, uint32_t slot) {
    const uint32_t i = ((slot * 0x9E3779B1u) >> 16) & (uint32_t)(kAggSize - 1);
    uint32_t cur = *VF_LDS(uint32_t, &a.slot[i]);
    if (cur == kNoSlot) { cur = atomicCAS(&a.slot[i], kNoSlot, slot); if (cur == kNoSlot) cur = slot; }
    if (cur != slot) return false;
    atomicAdd(&a.cnt[i], 1u);
    return true;
}

// ---- the count log as k_finish writes it (CountLog, vf_device.h) ---------------------------------
// Per wave: a segment of slot indices, appended in place with one store per trip - the lanes that log write consecutive
// dwords, the line fills up in L2 over the next trips and leaves it once.  Per workgroup: the (slot, count) pairs of the
// aggregator and the key cache at kernel end, and the number of entries per slot part in LDS (s_tot), added to the
// batch's totals when the workgroup is done: k_log_scatter turns the totals into part starts.
struct WaveLog {
    uint32_t *seg;                           // this wave's segment of log.one
    uint32_t n;                              // entries written (wave-uniform)
};
__device__ __forceinline__ void log_one(WaveLog &wl, const CountLog &log, uint32_t *s_tot, uint32_t slot, bool want) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    if (want) {
        const uint32_t at = wl.n + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        if (at < log.cap_one) wl.seg[at] = slot;           // (cap_one = the reads a wave can see: never exceeded)
        if (log.parts) atomicAdd(&s_tot[slot >> log.shift], 1u);
    }
    wl.n += (uint32_t)__popcll(m);
}
__device__ __forceinline__ void log_many(const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany, uint32_t slot, uint32_t cnt) {
    const uint32_t at = atomicAdd(s_nmany, 1u);
    if (at < kLogMany) log.many[(uint64_t)blockIdx.x * kLogMany + at] = make_uint2(slot, cnt);
    if (log.parts) atomicAdd(&s_tot[kLogMaxParts + (slot >> log.shift)], 1u);
}
__device__ __forceinline__ void agg_flush(Agg &a, const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany) {
    for (int i = threadIdx.x; i < kAggSize; i += kBlock)
        if (a.slot[i] != kNoSlot && a.cnt[i]) log_many(log, s_tot, s_nmany, a.slot[i], a.cnt[i]);
}

// Workgroup-level key cache: the occurrences of a workgroup's commonest keys, counted in LDS by key, in front of the table.
// The first lane of the workgroup that meets a key while the key's entry is free claims the entry, then takes the table
// path itself (lookup, insert queue) and publishes the slot it ends up with; every later read of that key in the
// workgroup - in the same wave a moment later, or ten thousand reads on - is counted in the entry and asks the table
// nothing.  What it saves is requests to ONE line: the first key of the Zipf stream is 12 % of the reads, the first ten
// a third, and every table lookup of a key is two or three 16-byte requests to the same L2 line from every CU of the
// XCD, served one after the other by that line's channel.  In a cleared table that was the launch's first 100 us: every
// wave's first insert batch held eight lanes of the commonest key and all 3 072 waves ran it at once (rounds 1-2 entered
// a key only once the table had it, i.e. after that storm).  Entry = 16 dwords:
//   { state, h1, h2, meta | slot, count, -, - | key[0..3] | key[4..7] }
//   state 0 free, 1 claimed (key and tags written by the claiming lane right after the claim; slot not yet known),
//   2 slot known.  The last tag word (meta) is release-stored after everything else of the entry and acquire-loaded first by
//   readers (keycache_match), so a reader that sees it sees the whole key; an entry is never replaced; meta is never 0 (a
//   counted key has a length).
// Direct mapped by h1; keys of more than eight dwords are not cached.  Flushed at kernel end as (slot, count) pairs.
#ifndef VF_KEYCACHE_SIZE
#define VF_KEYCACHE_SIZE 256
#endif
constexpr int kKeyCacheSize = VF_KEYCACHE_SIZE;        // power of two
struct KeyCache { u32x4 q[kKeyCacheSize][4]; };
__device__ __forceinline__ void keycache_init(KeyCache &kc) {
    for (int i = threadIdx.x; i < kKeyCacheSize; i += kBlock) {
#pragma unroll
        for (int q = 0; q < 4; ++q) kc.q[i][q] = u32x4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ uint32_t keycache_index(uint32_t h1) { return ((h1 * 0x9E3779B1u) >> 16) & (uint32_t)(kKeyCacheSize - 1); }
// Publication of an entry's key (C++ / HSA memory model, workgroup scope, LDS): the claiming lane writes the key words, the
// slot word and two of the three tag words with plain stores and then RELEASE-stores the last tag word, `meta` (never 0; 0
// until then - entries are never replaced); a reader ACQUIRE-loads `meta` first and looks at the other words only behind a
// match.  Rounds 2-3 wrote tags after key words with plain volatile stores and leaned on one CU's LDS executing the DS
// operations of different waves in arrival order - true of the hardware, not of the memory model (a false hit would count a
// read against the wrong key's slot, silently).  On gfx950 the release costs one s_waitcnt lgkmcnt(0) in the claiming lane's
// wave, once per entry; the acquire is the wait the compare needs anyway.
__device__ __forceinline__ uint32_t keycache_meta_acquire(KeyCache &kc, uint32_t i) {
    return __hip_atomic_load(reinterpret_cast<uint32_t *>(&kc.q[i][0]) + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool keycache_match(KeyCache &kc, uint32_t i, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords]) {
    if (keycache_meta_acquire(kc, i) != meta) return false;
    const u32x4 t0 = *VF_LDS(u32x4, &kc.q[i][0]);
    const u32x4 k0 = *VF_LDS(u32x4, &kc.q[i][2]), k1 = *VF_LDS(u32x4, &kc.q[i][3]);
    const uint32_t diff = (t0.y ^ h1) | (t0.z ^ h2) | (k0.x ^ kin[0]) | (k0.y ^ kin[1]) | (k0.z ^ kin[2]) | (k0.w ^ kin[3]) |
                          (k1.x ^ kin[4]) | (k1.y ^ kin[5]) | (k1.z ^ kin[6]) | (k1.w ^ kin[7]);
    return diff == 0u;
}
// What the cache does with one read of a key: kCacheCounted = counted in the key's entry, done; kCacheOwner = this lane
// has just claimed the entry for the key: it goes on to the table (where its own occurrence is counted like any other) and
// publishes the slot; kCacheNone = the entry belongs to another key (or the key is not cacheable): table path.
enum { kCacheNone = 0, kCacheCounted = 1, kCacheOwner = 2 };
__device__ __forceinline__ int keycache_probe(KeyCache &kc, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords], bool can) {
    const uint32_t i = keycache_index(h1);
    uint32_t *st = reinterpret_cast<uint32_t *>(&kc.q[i][0]);
    int role = kCacheNone;
    if (can && *VF_LDS(uint32_t, st) == 0u && atomicCAS(st, 0u, 1u) == 0u) {
        *VF_LDS(u32x4, &kc.q[i][2]) = u32x4{kin[0], kin[1], kin[2], kin[3]};
        *VF_LDS(u32x4, &kc.q[i][3]) = u32x4{kin[4], kin[5], kin[6], kin[7]};
        *VF_LDS(u32x4, &kc.q[i][1]) = u32x4{kNoSlot, 0u, 0u, 0u};
        VF_LDS(uint32_t, st)[1] = h1; VF_LDS(uint32_t, st)[2] = h2;
        __hip_atomic_store(st + 3, meta, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        role = kCacheOwner;
    }
    // (the lanes of this wave that lost the claim to a lane with the same key find the key published: the release store
    // above is earlier in the same instruction stream)
    if (can && role == kCacheNone && keycache_match(kc, i, h1, h2, meta, kin)) {
        atomicAdd(reinterpret_cast<uint32_t *>(&kc.q[i][1]) + 1, 1u);
        role = kCacheCounted;
    }
    return role;
}
// the table slot of a key (found or inserted, by whichever lane of the workgroup got there): noted in the key's entry if
// it has one that still waits for it.  Two lanes may both note a slot - each is a slot of that key (duplicates of a key
// in the table are summed when the result is built), either will do.
__device__ __forceinline__ void keycache_publish(KeyCache &kc, uint32_t h1, uint32_t h2, uint32_t meta, const uint32_t (&kin)[kInlineDwords], uint32_t slot, bool can) {
    if (!can || slot == kNoSlot) return;
    const uint32_t i = keycache_index(h1);
    volatile __attribute__((address_space(3))) uint32_t *st = VF_LDS(uint32_t, &kc.q[i][0]);
    if (st[0] != 1u || !keycache_match(kc, i, h1, h2, meta, kin)) return;
    VF_LDS(uint32_t, &kc.q[i][1])[0] = slot;
    // (slot and state are read back only by keycache_flush, behind the workgroup barrier at kernel end; the release keeps
    // the pair ordered for whoever reads them earlier)
    __hip_atomic_store(reinterpret_cast<uint32_t *>(&kc.q[i][0]), 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// after the workgroup's last table access.  An entry with occurrences but no slot means its owner's insert failed: the
// table or the arena is full, which the insert has flagged; flagged again here so that it cannot pass unseen.
__device__ __forceinline__ void keycache_flush(KeyCache &kc, const Table &t, const CountLog &log, uint32_t *s_tot, uint32_t *s_nmany) {
    for (int i = threadIdx.x; i < kKeyCacheSize; i += kBlock) {
        const u32x4 t0 = kc.q[i][0], t1 = kc.q[i][1];
        if (t0.x == 2u && t1.y && VF_BOUND(t, t1.x <= t.slot_mask, 8, t1.x, t1.y)) log_many(log, s_tot, s_nmany, t1.x, t1.y);
        else if (t0.x == 1u && t1.y) atomicOr((unsigned long long *)&t.counters[C_ERROR], 16ull);
    }
}


}  // namespace vf
