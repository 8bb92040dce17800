// Guard allocator for PyTorch-ROCm (torch.cuda.memory.CUDAPluggableAllocator) -- TEST INFRASTRUCTURE, not part of the product.
//
// The product hands raw device pointers to asynchronous launches and bakes them into launch plans and hipGraphs; PyTorch's caching
// allocator hides every mistake of that kind (freed blocks stay mapped and are recycled, neighbours of a tensor are always mapped).
// This allocator makes them visible (tools/guard_run.py, tests/test_gpu_guard.py):
//
//   mode "vmm" (default)  every allocation is its OWN virtual-memory mapping (hipMemAddressReserve / hipMemCreate / hipMemMap):
//       * the tensor ends at the END of its mapping and the address range behind it stays unmapped: a read or write past the end of
//         an operand is a GPU memory fault at once, whatever the values are used for (tile loads of padding rows that are discarded
//         later are exactly the accesses that fault one day when the tensor happens to end a segment);
//       * free = wait for the device (so no queued launch is cut off -- this tool does not look for missing stream ordering),
//         then UNMAP at once and never hand the address range out again: any LATER launch through a pointer to a freed tensor
//         (a plan / hipGraph that outlived its operands) faults instead of reading recycled memory;
//       * the slack in front of the tensor (mapping granularity) holds a canary pattern, checked on free: writes BEFORE the start.
//   mode "canary"  hipMalloc with canary zones on both sides, checked on free and by guard_check(): out-of-bounds WRITES without
//       any fault (a run that must not kill the process).
//   fresh memory is filled with a NaN pattern in both modes: reading memory nobody wrote shows up as non-finite results.
//
// Allocations made while a stream capture is active belong to the captured graph (it replays their addresses): they are never
// freed, and frees of other blocks that arrive during a capture are deferred until the capture is over.
//
//   build: hipcc -O2 -fPIC -shared tools/guard_alloc.cpp -o tools/_build/libdsc_guard_alloc.so   (__graft_entry__.build() does it)
//   use  : torch.cuda.memory.change_current_allocator(CUDAPluggableAllocator(so, "guard_malloc", "guard_free"))  BEFORE the first
//          device allocation of the process
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

struct Block {
    void* va = nullptr;          // vmm: start of the reservation;  canary: hipMalloc pointer
    size_t va_size = 0;          // vmm: reservation (mapping + unmapped gap)
    size_t map_size = 0;         // vmm: mapped bytes;  canary: total bytes
    hipMemGenericAllocationHandle_t handle{};
    size_t user = 0;             // bytes the caller asked for
    size_t front = 0;            // bytes between the start of the mapping and the user pointer (canary)
    size_t back = 0;             // canary mode: bytes behind the user range
    bool in_capture = false;
    bool vmm = false;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
std::vector<void*> g_deferred;
int g_mode = -1;                 // 0 vmm, 1 canary
size_t g_gran = 0, g_gap = 0, g_zone = 4096, g_align = 16;
bool g_poison = true, g_verbose = false;
long g_allocs = 0, g_frees = 0, g_capture_allocs = 0, g_deferred_frees = 0, g_corrupt = 0;
size_t g_live_bytes = 0, g_peak_bytes = 0, g_va_reserved = 0;

constexpr unsigned char CANARY = 0xA5;
constexpr uint32_t POISON = 0x7fc0dead;        // a quiet NaN as f32; a huge value as int64

#define GCHECK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            fprintf(stderr, "guard_alloc: %s -> %s\n", #expr, hipGetErrorString(e__));                 \
            abort();                                                                                   \
        }                                                                                              \
    } while (0)

// HIP calls that are illegal while ANOTHER thread-global capture is running are allowed in relaxed mode
struct Relaxed {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    Relaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    ~Relaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

void init_locked(int device) {
    if (g_mode >= 0) return;
    const char* m = getenv("DSC_GUARD_MODE");
    g_mode = (m && !strcmp(m, "canary")) ? 1 : 0;
    if (const char* p = getenv("DSC_GUARD_POISON")) g_poison = atoi(p) != 0;
    if (const char* p = getenv("DSC_GUARD_ALIGN")) g_align = (size_t)atol(p);
    if (const char* p = getenv("DSC_GUARD_VERBOSE")) g_verbose = atoi(p) != 0;
    if (g_align < 16 || (g_align & (g_align - 1))) g_align = 16;
    if (g_mode == 0) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
            fprintf(stderr, "guard_alloc: the virtual-memory API is not available here: falling back to canary mode\n");
            g_mode = 1;
        } else {
            g_gran = gran;
            g_gap = gran > (size_t)(2 << 20) ? gran : (size_t)(2 << 20);      // unmapped range behind every tensor
            g_gap = (g_gap + gran - 1) / gran * gran;
        }
    }
    fprintf(stderr, "guard_alloc: mode %s, granularity %zu bytes, gap %zu bytes, alignment %zu, poison %d\n", g_mode ? "canary" : "vmm",
            g_gran, g_gap, g_align, (int)g_poison);
}

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return true;                 // the legacy stream cannot be queried while another stream captures: treat as "a capture is on"
    }
    return st != hipStreamCaptureStatusNone;
}

bool check_canary_locked(const Block& b, void* user, const char* when) {
    // device -> host copies of the zones (the caller has synchronised)
    bool ok = true;
    std::vector<unsigned char> host;
    auto scan = [&](const char* which, const unsigned char* dev, size_t n) {
        if (!n) return;
        host.resize(n);
        GCHECK(hipMemcpy(host.data(), dev, n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
            if (host[i] != CANARY) {
                fprintf(stderr, "guard_alloc: CANARY OVERWRITTEN %s the tensor at %p (%zu bytes), zone offset %zu of %zu, found 0x%02x (%s)\n",
                        which, user, b.user, i, n, host[i], when);
                ok = false;
                break;
            }
    };
    scan("in front of", (const unsigned char*)user - b.front, b.front);
    if (b.back) scan("behind", (const unsigned char*)user + b.user, b.back);
    if (!ok) ++g_corrupt;
    return ok;
}

void release_locked(void* user, const Block& b) {
    check_canary_locked(b, user, "free");
    if (b.vmm) {
        void* map0 = (char*)b.va;
        GCHECK(hipMemUnmap(map0, b.map_size));
        GCHECK(hipMemRelease(b.handle));
        // the address range is NOT returned (hipMemAddressFree): a stale pointer must keep faulting, never land in a newer tensor
    } else {
        GCHECK(hipFree(b.va));
    }
    g_live_bytes -= b.user;
    ++g_frees;
}

void drain_deferred_locked() {
    if (g_deferred.empty()) return;
    GCHECK(hipDeviceSynchronize());
    for (void* p : g_deferred) {
        auto it = g_live.find(p);
        if (it == g_live.end()) continue;
        release_locked(p, it->second);
        g_live.erase(it);
    }
    g_deferred.clear();
}

}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    Relaxed relaxed;
    init_locked(device);
    const bool cap = capturing(stream);
    if (!cap) drain_deferred_locked();
    Block b;
    b.user = (size_t)size;
    b.in_capture = cap;
    void* user = nullptr;
    const size_t padded = (b.user + g_align - 1) / g_align * g_align;
    if (g_mode == 0) {
        b.vmm = true;
        b.map_size = (padded + g_gran - 1) / g_gran * g_gran;
        b.va_size = b.map_size + g_gap;
        GCHECK(hipMemAddressReserve(&b.va, b.va_size, g_gran, nullptr, 0));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        GCHECK(hipMemCreate(&b.handle, b.map_size, &prop, 0));
        GCHECK(hipMemMap(b.va, b.map_size, 0, b.handle, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        GCHECK(hipMemSetAccess(b.va, b.map_size, &acc, 1));
        b.front = b.map_size - padded;                 // the tensor's (padded) end is the end of the mapping
        user = (char*)b.va + b.front;
        g_va_reserved += b.va_size;
    } else {
        b.front = g_zone;
        b.back = g_zone + (padded - b.user);
        b.map_size = b.front + b.user + b.back;
        GCHECK(hipMalloc(&b.va, b.map_size));
        user = (char*)b.va + b.front;
    }
    if (!cap) {
        // canary zones + poison, ordered on the caller's stream (the tensor is used on it); fresh memory holds NaNs
        if (b.front) GCHECK(hipMemsetAsync((char*)user - b.front, CANARY, b.front, stream));
        if (b.back) GCHECK(hipMemsetAsync((char*)user + b.user, CANARY, b.back, stream));
        if (g_poison && b.user >= 4) GCHECK(hipMemsetD32Async((hipDeviceptr_t)user, (int)POISON, b.user / 4, stream));
    } else {
        b.front = 0;                                   // nothing may be enqueued on a capturing stream on the graph's behalf
        b.back = 0;
        ++g_capture_allocs;
    }
    g_live[user] = b;
    ++g_allocs;
    g_live_bytes += b.user;
    if (g_live_bytes > g_peak_bytes) g_peak_bytes = g_live_bytes;
    if (g_verbose) fprintf(stderr, "guard_alloc: + %p %zu bytes%s\n", user, b.user, cap ? " (capture)" : "");
    return user;
}

extern "C" void guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device;
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Relaxed relaxed;
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        fprintf(stderr, "guard_alloc: free of an unknown pointer %p\n", ptr);
        return;
    }
    if (it->second.in_capture) return;                 // owned by a captured graph: its replays use the address
    if (capturing(stream)) {
        g_deferred.push_back(ptr);
        ++g_deferred_frees;
        return;
    }
    drain_deferred_locked();
    GCHECK(hipDeviceSynchronize());                    // no queued launch loses an operand to this free
    if (g_verbose) fprintf(stderr, "guard_alloc: - %p %zu bytes\n", ptr, it->second.user);
    release_locked(ptr, it->second);
    g_live.erase(it);
}

// Check the canary zones of every live tensor (synchronises).  Returns the number of corrupted zones seen so far (frees included).
extern "C" long guard_check(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    Relaxed relaxed;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    for (auto& kv : g_live)
        if (!kv.second.in_capture) check_canary_locked(kv.second, kv.first, "guard_check");
    return g_corrupt;
}

extern "C" void guard_stats(long* out /* [8] */) {
    std::lock_guard<std::mutex> lk(g_mu);
    out[0] = g_allocs; out[1] = g_frees; out[2] = (long)g_live.size(); out[3] = (long)(g_live_bytes >> 20);
    out[4] = (long)(g_peak_bytes >> 20); out[5] = g_capture_allocs; out[6] = g_deferred_frees; out[7] = g_corrupt;
}

extern "C" int guard_mode(void) { return g_mode; }
