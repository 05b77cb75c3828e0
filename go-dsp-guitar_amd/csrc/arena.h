/* arena.h -- sub-allocator for the device memory of the per-unit state (ctx.h, api_plan.cpp).  Written against a small backend (malloc / free /
 * fill / wait) so that the book-keeping can be exercised on the host: tests/native/arena_check.cpp runs it over plain memory. */
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <iterator>
#include <map>
#include <utility>
#include <vector>

/* Device memory of the per-unit state.  A 512-channel context owns ~17 000 blocks (per unit: small state, history ring, and per
 * power amp the overlap-save history, delay line, product spectra, frame counter, IR spectra); as one hipMalloc / hipFree each they
 * cost ~1 s to release and scatter the state over the address space.  They come out of a few chunks instead (the first 4 MiB, every
 * further one as large as everything before it, up to 1 GiB -- a one-channel context holds 4 MiB, a 512-channel one a dozen chunks),
 * sub-allocated on the host: first fit over an address-ordered free list that coalesces on free.  No implicit synchronisation: whoever
 * frees a block has already waited for the work that used it (every call site does; GDG_ARENA_SYNC_RELEASE=1 synchronises the device in
 * release() to catch a call site that forgets to).
 * ZEROS.  Almost every block starts out as zeros (unit state, history rings, delay lines), and one hipMemsetAsync per block was 14 338
 * fill dispatches -- a third of the GPU time of a 512-channel context's set-up trace.  A chunk is zeroed ONCE, when it is made, and
 * remembers how far it has been handed out (`virgin`): alloc_zeroed() on space beyond that mark is free, only recycled space is filled. */
/* Chunks after the first that become entirely free are given back to the device, all but one (a run of 1M-tap filters does not pin its
 * gigabytes for the life of the context); the first chunk stays.  WHEN: freeing device memory waits for the whole device, so a context
 * that serves a live stream must not do it inside a process call.  With `defer_trim` (the device arena of the api_*.cpp files) release() only notes
 * that there is something to give back and trim() does it -- api_plan.cpp / api_process.cpp call trim() where it has just drained the stream anyway (a plan
 * rebuild, gdg_ctx_trim, the context's end), never on the parameter-patch path of a process call.  What CAN still stall a patch: a history
 * that outgrows every hole makes a new chunk (one device malloc + one fill of up to 1 GiB, waited for); gdg.h says so at
 * gdg_unit_set_param. */
template <class B>
struct ArenaT {
    using err_t = typename B::err_t;
    using stream_t = typename B::stream_t;
    struct Chunk { char *base; size_t size; std::map<size_t, size_t> holes; size_t virgin; };   /* holes: offset -> bytes; [virgin, size) was never handed out: zeros */
    std::vector<Chunk> chunks;                                                         /* a trimmed chunk keeps its slot with base == nullptr */
    std::map<const void *, std::pair<size_t, size_t>> live;                           /* block -> (chunk index, bytes) */
    size_t total = 0, peak_total = 0;
    stream_t stream = stream_t();                                                      /* the context's stream: chunk zeroing and fills are ordered on it */
    size_t fills = 0, fills_saved = 0;                                                 /* alloc_zeroed: fill dispatches issued / avoided */
    size_t trimmed = 0;                                                                /* chunks given back */
    /* Blocks of a page or more start on `big_align` (env GDG_ARENA_ALIGN, default 4 KiB like a hipMalloc of their own would): the
     * streaming kernels read delay lines and spectra front to back, and packing those at 256-byte offsets behind the small state
     * blocks cost the convolution 5-13 % (profiles/arena_ab_r03.txt).  GDG_ARENA=0: one hipMalloc per block (A/B measurements). */
    size_t big_align = 4096;
    size_t first_chunk = (size_t)4 << 20;
    bool direct = false, sync_release = false;
    bool defer_trim = false, trim_pending = false;
    size_t spare_hint = 0;                                                             /* the chunk that became free last: the spare trim() keeps */
    ArenaT() {
        if (const char *e = getenv("GDG_ARENA_ALIGN")) { size_t a = (size_t)atoll(e); if (a >= 256 && (a & (a - 1)) == 0) big_align = a; }
        if (const char *e = getenv("GDG_ARENA")) direct = atoi(e) == 0;
        if (const char *e = getenv("GDG_ARENA_FIRST_CHUNK")) { size_t a = (size_t)atoll(e); if (a >= 4096) first_chunk = a; }
        if (const char *e = getenv("GDG_ARENA_SYNC_RELEASE")) sync_release = atoi(e) != 0;
    }
    static size_t round_up(size_t b, size_t a) { return (b + a - 1) & ~(a - 1); }
    /* *zeroed (optional): the block is known to hold zeros (never handed out since its chunk was made) */
    err_t alloc(void **out, size_t bytes, bool *zeroed = nullptr) {
        if (zeroed) *zeroed = false;
        if (direct) return B::malloc(out, bytes ? bytes : 1);
        const size_t need = round_up(bytes ? bytes : 1, 256);
        const size_t align = need >= 4096 ? big_align : 256;
        for (size_t c = 0; c < chunks.size(); c++) {
            auto &h = chunks[c].holes;
            const uintptr_t base = (uintptr_t)chunks[c].base;
            for (auto it = h.begin(); it != h.end(); ++it) {
                const size_t off = it->first, end = off + it->second;
                const size_t at = (size_t)(round_up(base + off, align) - base);
                if (at + need > end) continue;
                h.erase(it);
                if (at > off) h.emplace(off, at - off);
                if (end > at + need) h.emplace(at + need, end - (at + need));
                *out = chunks[c].base + at;
                live.emplace(*out, std::make_pair(c, need));
                /* zeros only if the WHOLE block lies in never-used space; the mark always moves past what is handed out (a freed block that
                 * coalesced with the untouched tail gives a hole that straddles the mark) */
                if (at >= chunks[c].virgin && zeroed) *zeroed = true;
                chunks[c].virgin = std::max(chunks[c].virgin, at + need);
                return B::ok();
            }
        }
        size_t size = std::max(need, std::min((size_t)1 << 30, std::max(first_chunk, total)));
        void *base = nullptr;
        err_t e = B::malloc(&base, size);
        if (e != B::ok() && size > need) { size = need; e = B::malloc(&base, size); }
        if (e != B::ok()) { *out = nullptr; return e; }
        /* zeros, once: ONE fill for everything this chunk will ever hand out for the first time.  Waited for: not every later writer
         * of the chunk is ordered on `stream` (synchronous copies of tables run on the null stream). */
        e = B::fill_zero(base, size, stream);
        if (e == B::ok()) e = B::wait(stream);
        if (e != B::ok()) { B::free(base); *out = nullptr; return e; }
        total += size;
        peak_total = std::max(peak_total, total);
        size_t slot = chunks.size();
        for (size_t c = 0; c < chunks.size(); c++) if (!chunks[c].base) { slot = c; break; }
        if (slot == chunks.size()) chunks.emplace_back();
        chunks[slot] = Chunk{ static_cast<char *>(base), size, {}, need };       /* device chunks of >= 2 MiB start on 2 MiB */
        if (size > need) chunks[slot].holes.emplace(need, size - need);
        *out = base;
        live.emplace(*out, std::make_pair(slot, need));
        if (zeroed) *zeroed = true;
        return B::ok();
    }
    /* a block of zeros; a fill is enqueued on `st` only when the space has been used before */
    err_t alloc_zeroed(void **out, size_t bytes, stream_t st) {
        bool zeroed = false;
        err_t e = alloc(out, bytes, &zeroed);
        if (e != B::ok()) return e;
        if (zeroed) { fills_saved++; return B::ok(); }
        fills++;
        return B::fill_zero(*out, bytes ? bytes : 1, st);
    }
    void release(const void *p) {
        if (!p) return;
        if (sync_release) B::wait_device();
        if (direct) { B::free(const_cast<void *>(p)); return; }
        auto it = live.find(p);
        if (it == live.end()) return;
        const size_t c = it->second.first;
        Chunk &ch = chunks[c];
        size_t off = (size_t)(static_cast<const char *>(p) - ch.base), n = it->second.second;
        live.erase(it);
        auto next = ch.holes.lower_bound(off);
        if (next != ch.holes.end() && off + n == next->first) { n += next->second; next = ch.holes.erase(next); }
        bool merged = false;
        if (next != ch.holes.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == off) { prev->second += n; merged = true; }
        }
        if (!merged) ch.holes.emplace(off, n);
        /* this chunk is now entirely free: it stays as the ONE spare (a temporary that lives alone in a chunk must not cost a device
         * malloc + free per use); an older spare goes back to the device */
        if (c > 0 && entirely_free(ch)) {
            if (defer_trim) { trim_pending = true; spare_hint = c; return; }
            give_back_all_but(c);
        }
    }
    /* give the entirely free chunks back, all but `keep` (the spare) */
    void give_back_all_but(size_t keep) {
        for (size_t d = 1; d < chunks.size(); d++) {
            if (d == keep || !entirely_free(chunks[d])) continue;
            B::free(chunks[d].base);                                               /* waits for the device, like any free of device memory */
            total -= chunks[d].size;
            trimmed++;
            chunks[d] = Chunk{ nullptr, 0, {}, 0 };
        }
    }
    /* deferred trimming: call where waiting for the device costs nothing (the caller has just drained it).  Keeps the chunk that became
     * free last (what the immediate mode keeps), or, if that one has been taken again, the first free one it finds. */
    void trim() {
        if (!trim_pending) return;
        trim_pending = false;
        size_t keep = (spare_hint > 0 && spare_hint < chunks.size() && entirely_free(chunks[spare_hint])) ? spare_hint : 0;
        for (size_t d = 1; d < chunks.size() && keep == 0; d++) if (entirely_free(chunks[d])) keep = d;
        if (keep) give_back_all_but(keep);
    }
    static bool entirely_free(const Chunk &ch) {
        return ch.base && ch.holes.size() == 1 && ch.holes.begin()->first == 0 && ch.holes.begin()->second == ch.size;
    }
    size_t chunks_held() const { size_t n = 0; for (auto &c : chunks) n += c.base != nullptr; return n; }
    void destroy() {
        for (auto &c : chunks) if (c.base) B::free(c.base);
        chunks.clear();
        live.clear();
        total = 0;
    }
};
