// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so that lets SEVERAL RANKS SHARE ONE GPU.
//
// RCCL refuses two ranks on one device, and this repository's GPU box has one: the N > 1 branch of the library's exchange
// (voxel-rs_amd/csrc/hip/comm.cpp: vx_comm_init, the grouped ncclRecv loop of vx_gather_tiles, the tickets, rank 0's assembly behind it) could
// never run. This double implements the eight entry points comm.cpp binds -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart,
// ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString -- between PROCESSES ON DEVICE 0: a POSIX shared-memory block carries the rendezvous and the
// flags (mapped into every rank's GPU address space), every rank exports a staging buffer through a HIP IPC memory handle, and a message travels as
//   sender, on the caller's stream:   [wait until the slot's previous message was consumed; copy sendbuf -> own staging slot] [flag sent = n]
//   receiver, on the caller's stream: [wait for sent >= n; copy the peer's staging slot -> recvbuf]                        [flag consumed = n]
// -- kernels that spin on flags, stream-ordered and asynchronous to the host like RCCL's own (ncclGroupEnd returns at once; a peer that never
// joins leaves a kernel waiting, which gives up after kGiveUpSeconds and raises the communicator's error word instead of hanging the GPU).
// It says nothing about RCCL; it is what lets tests/test_multirank_one_gpu.py execute the library's and bench.py's multi-rank code paths.
// Loaded only through vx_comm_library(path) by that test; nothing of the product links or ships it.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kSlots = 2;                        // staging slots per (sender, receiver) pair
constexpr size_t kSlotBytes = size_t(40) << 20;  // a 1080p RGBA32F frame's half (two ranks) is 16.6 MB
constexpr unsigned kGiveUpSeconds = 8;
constexpr uint32_t kCopyGroups = 64;             // workgroups of the copy kernels (256 threads each)

struct Shared {  // the shared-memory block (zero-filled by its creator)
    uint32_t ready[kMaxRanks];                    // rank r has published its handle
    hipIpcMemHandle_t staging[kMaxRanks];         // rank r's staging buffer: [receiver][slot][kSlotBytes]
    uint32_t sent[kMaxRanks][kMaxRanks];          // [sender][receiver]: messages the sender has put into its staging
    uint32_t consumed[kMaxRanks][kMaxRanks];      // [sender][receiver]: messages the receiver has copied out
    uint32_t error[kMaxRanks];                    // a kernel of rank r gave up waiting
    uint32_t left[kMaxRanks];                     // rank r has destroyed its communicator
};

}  // namespace

struct ncclComm {
    int nranks = 0, rank = 0;
    std::string shm_name;
    Shared* host = nullptr;     // the block, as this process sees it
    Shared* dev = nullptr;      // ... and as its GPU does (hipHostRegister)
    uint8_t* staging = nullptr; // this rank's
    uint8_t* peer[kMaxRanks] = {};  // the other ranks' staging buffers, opened through their IPC handles
    uint32_t n_sent[kMaxRanks] = {}, n_received[kMaxRanks] = {};
};

namespace {

__device__ __forceinline__ uint32_t load_flag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// every workgroup waits for *flag >= want (the message is there / the slot is free), then copies its share
__global__ __launch_bounds__(256) void wait_and_copy(const uint32_t* flag, uint32_t want, uint32_t* error, const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                     const uint8_t* src_tail, uint8_t* dst_tail, uint32_t tail) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (int32_t(load_flag(flag) - want) < 0) {
            __builtin_amdgcn_s_sleep(64);
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)kGiveUpSeconds * 100000000ull) {  // (100 MHz)
                ok = 0;
                __hip_atomic_store(error, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    if (!ok) return;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

__global__ void set_flag(uint32_t* flag, uint32_t value) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

uint8_t* slot_of(uint8_t* staging, int receiver, uint32_t n) { return staging + (size_t(receiver) * kSlots + (n - 1u) % kSlots) * kSlotBytes; }

ncclResult_t enqueue(hipStream_t stream, const uint32_t* flag, uint32_t want, uint32_t* error, const void* src, void* dst, size_t bytes, uint32_t* done_flag, uint32_t done_value) {
    const size_t n16 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) ? 0 : bytes / 16;
    const size_t tail = bytes - n16 * 16;
    if (tail > 255) {  // (unaligned buffers: the library's tile lists never are)
        fprintf(stderr, "stub rccl: buffers must be 16-byte aligned\n");
        return ncclInvalidArgument;
    }
    hipLaunchKernelGGL(wait_and_copy, dim3(kCopyGroups), dim3(256), 0, stream, flag, want, error, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16,
                       static_cast<const uint8_t*>(src) + n16 * 16, static_cast<uint8_t*>(dst) + n16 * 16, uint32_t(tail));
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, stream, done_flag, done_value);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    unsigned long long r[2] = {(unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(), (unsigned long long)getpid()};
    if (FILE* f = fopen("/dev/urandom", "rb")) {
        if (fread(r, sizeof r, 1, f) != 1) r[1] ^= 0x9e3779b97f4a7c15ull;
        fclose(f);
    }
    snprintf(id->internal, sizeof id->internal, "/vxstub_rccl_%016llx%016llx", r[0], r[1]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    ncclComm* c = new ncclComm();
    c->nranks = nranks;
    c->rank = rank;
    c->shm_name = id.internal;
    const int fd = shm_open(c->shm_name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->host = static_cast<Shared*>(m);
    if (hipHostRegister(m, sizeof(Shared), hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(reinterpret_cast<void**>(&c->dev), m, 0) != hipSuccess) {
        fprintf(stderr, "stub rccl: cannot map the shared block into the GPU: %s\n", hipGetErrorString(hipGetLastError()));
        delete c;
        return ncclUnhandledCudaError;
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->staging), size_t(nranks) * kSlots * kSlotBytes) != hipSuccess ||
        hipIpcGetMemHandle(&c->host->staging[rank], c->staging) != hipSuccess) {
        fprintf(stderr, "stub rccl: staging buffer: %s\n", hipGetErrorString(hipGetLastError()));
        delete c;
        return ncclUnhandledCudaError;
    }
    __atomic_store_n(&c->host->ready[rank], 1u, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nranks; ++r) {
        while (__atomic_load_n(&c->host->ready[r], __ATOMIC_ACQUIRE) == 0u) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                fprintf(stderr, "stub rccl: rank %d never arrived\n", r);
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (r == rank) continue;
        if (hipIpcOpenMemHandle(reinterpret_cast<void**>(&c->peer[r]), c->host->staging[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            fprintf(stderr, "stub rccl: hipIpcOpenMemHandle(rank %d): %s\n", r, hipGetErrorString(hipGetLastError()));
            return ncclUnhandledCudaError;
        }
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->nranks; ++r)
        if (c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    // a peer may still be copying out of this rank's staging buffer: it goes when everybody has left (or after a grace period)
    __atomic_store_n(&c->host->left[c->rank], 1u, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->nranks; ++r)
        while (__atomic_load_n(&c->host->left[r], __ATOMIC_ACQUIRE) == 0u && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5))
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (c->staging) (void)hipFree(c->staging);
    (void)hipHostUnregister(c->host);
    munmap(c->host, sizeof(Shared));
    if (c->rank == 0) shm_unlink(c->shm_name.c_str());
    delete c;
    return ncclSuccess;
}

// (every operation is enqueued where it is called; the group calls only bracket them)
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    const size_t bytes = count * type_bytes(type);
    if (!c || !sendbuff || peer < 0 || peer >= c->nranks || peer == c->rank || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
    const uint32_t n = ++c->n_sent[peer];
    // the slot's previous message (n - kSlots) must have been copied out; then sendbuf -> slot; then sent = n
    return enqueue(stream, &c->dev->consumed[c->rank][peer], n > uint32_t(kSlots) ? n - uint32_t(kSlots) : 0u, &c->dev->error[c->rank], sendbuff,
                   slot_of(c->staging, peer, n), bytes, &c->dev->sent[c->rank][peer], n);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    const size_t bytes = count * type_bytes(type);
    if (!c || !recvbuff || peer < 0 || peer >= c->nranks || peer == c->rank || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
    const uint32_t n = ++c->n_received[peer];
    // the peer's n-th message to this rank must be in its staging; then slot -> recvbuf; then consumed = n
    return enqueue(stream, &c->dev->sent[peer][c->rank], n, &c->dev->error[c->rank], slot_of(c->peer[peer], c->rank, n), recvbuff, bytes,
                   &c->dev->consumed[peer][c->rank], n);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (stub rccl)";
        case ncclUnhandledCudaError: return "unhandled HIP error (stub rccl)";
        case ncclSystemError: return "system error (stub rccl)";
        case ncclInvalidArgument: return "invalid argument (stub rccl)";
        default: return "error (stub rccl)";
    }
}

}  // extern "C"
