// pt_debug_hooks.hpp — everything the diagnostic builds compile INTO the kernels, in one place; all of it is compiled out of the
// product library (no flag: empty macros).
//   -DPT_AUDIT    every read-modify-write of an accumulation pixel mirrored by a device-scope atomic side word (tools/handover_stress.cpp)
//   -DPT_CHAOS    pseudo-random s_sleep delays at the hand-over protocol's decision points
//   -DPT_PROFILE  per-section cycle counters of the bounce iteration (tools/profile_sections.py)
#pragma once
#include "pt_kernels.hpp"
#include "pt_math.hpp"

// Section profiling (hipcc -DPT_PROFILE): per-wavefront s_memtime deltas accumulated per section of the bounce iteration and added to
// FrameArgs::timeline[200000..] at the end of the kernel.
#ifdef PT_PROFILE
#define PROF_PARAM , unsigned long long *prof
#define PROF_PASS , prof
#define PROF_DUMMY , prof_dummy
#define PROF_BEGIN unsigned long long prof_t = __builtin_readcyclecounter();
#define PROF_MARK(slot) { unsigned long long n_ = __builtin_readcyclecounter(); prof[slot] += n_ - prof_t; prof_t = n_; }
#else
#define PROF_PARAM
#define PROF_PASS
#define PROF_DUMMY
#define PROF_BEGIN
#define PROF_MARK(slot)
#endif

namespace pt {

// ---- hand-over audit and chaos injection (tools/handover_stress.cpp; compiled out of the product library).
// PT_AUDIT: every read-modify-write of an accumulation pixel (compute.glsl:126-129) is mirrored by ONE device-scope atomic
// exchange on a 64-bit side word per pixel: (frames folded so far) << 32 | hash(colour stored).  The exchange returns what the
// previous resolve of that pixel left there, so a resolve that ran out of order (frame f before f-1, or twice), or that folded
// into a colour other than the one the previous resolve stored (a stale or torn 16-byte read), is caught the moment it happens,
// independently of the alpha tags the product protocol relies on.  All-ones = history unknown (after a clear / reset / restore).
// PT_CHAOS: pseudo-random s_sleep delays (0.4 us ... 100 us) at the protocol's decision points, to widen every race window.
#ifdef PT_AUDIT
PT_DEV uint32_t audit_hash(float x, float y, float z)
{
    uint32_t a = __float_as_uint(x), b = __float_as_uint(y), c = __float_as_uint(z);
    uint32_t h = a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
    return h ^ (h >> 16);
}
// `p`: pixel index relative to a.accum; F: absolute frame being folded; `last`: the value that was loaded; `next`: the value
// about to be stored; site: which resolve site of which kernel (for the log)
PT_DEV void audit_resolve(const FrameArgs &a, size_t p, int F, float4 last, float4 next, int site)
{
    if (!a.audit) return;
    const unsigned long long now = ((unsigned long long)(uint32_t)(F + 1) << 32) | audit_hash(next.x, next.y, next.z);
    const unsigned long long old = atomicExch(a.audit + p, now);
    if (old == ~0ull) return;
    const uint32_t oldFrames = (uint32_t)(old >> 32), oldHash = (uint32_t)old, lastHash = audit_hash(last.x, last.y, last.z);
    if (oldFrames == (uint32_t)F && (F == 0 || oldHash == lastHash)) return;
    const unsigned int slot = atomicAdd(a.auditLog, 1u);
    if (slot >= (unsigned int)kAuditLogRecords) return;
    unsigned int *r = a.auditLog + 4 + slot * kAuditRecordWords;
    r[0] = (unsigned int)site | (oldFrames != (uint32_t)F ? 0x100u : 0u) | (oldHash != lastHash ? 0x200u : 0u);
    r[1] = (unsigned int)p;
    r[2] = (unsigned int)F;
    r[3] = oldFrames;
    r[4] = oldHash;
    r[5] = lastHash;
    r[6] = __float_as_uint(last.w);
    r[7] = a.launchSeq;
    r[8] = (unsigned int)a.frame | ((unsigned int)a.batchFrames << 24);
    r[9] = __float_as_uint(a.chainTag);
    r[10] = blockIdx.x;
    r[11] = (unsigned int)a.tagged | ((unsigned int)a.keepTags << 1) | ((unsigned int)a.variant << 8);
}
#define AUDIT_RESOLVE(a, p, F, last, next, site) audit_resolve(a, p, F, last, next, site)
#define AUDIT_SABOTAGED(a, pix, fj) ((a).auditSabotage > 0 && ((unsigned int)(pix) * 2654435761u + (unsigned int)(fj) * 40503u) % (unsigned int)(a).auditSabotage == 0u)
#else
#define AUDIT_RESOLVE(a, p, F, last, next, site)
#define AUDIT_SABOTAGED(a, pix, fj) false
#endif

#ifdef PT_CHAOS
// stateless: the wavefront's cycle counter hashed with the site; 3/4 of the calls do nothing, 3/16 sleep 0.4 - 3 us, 1/16 up to 100 us
PT_DEV void chaos_point(unsigned int site)
{
    uint32_t r = (uint32_t)__builtin_readcyclecounter();
    r = (r ^ (r >> 7)) * 0x9E3779B1u + site * 0x85EBCA6Bu;
    r ^= r >> 15;
    r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    if ((r & 3u) != 0u) return;
    const bool longNap = ((r >> 2) & 3u) == 0u;
    const int n = (int)((r >> 4) & (longNap ? 31u : 7u)) + 1;
    for (int i = 0; i < n; i++) {
        if (longNap) __builtin_amdgcn_s_sleep(127);
        else __builtin_amdgcn_s_sleep(16);
    }
}
#define CHAOS(site) chaos_point(site)
#else
#define CHAOS(site)
#endif

} // namespace pt
