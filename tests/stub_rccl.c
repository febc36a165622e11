/* stub_rccl.c — a stand-in for librccl that lets TWO ranks on ONE GPU drive the library's own transport
 * (rcw_comm_unique_id / rcw_comm_init / rcw_gather_*, include/rcw.h) end to end: test infrastructure only.
 *
 * Real RCCL refuses two ranks on one device, and this pipeline's boxes have one GPU; so the path
 * `rcw_comm_init(h, uid, rank = 1, world = 2)` and the uid hand-over in sharded.py could never execute.  librcw_hip
 * loads its collective library by name at run time (RCW_RCCL_LIBRARY, the one environment variable it reads); pointed
 * at this file's .so it calls the same eight entry points, which are implemented here with host shared memory:
 *   ncclGetUniqueId      128 random-ish bytes (pid, time, counter)
 *   ncclCommInitRank     opens /dev/shm/rcw_stub_<uid prefix> (rank 0 creates it), records (rank, world, uid) in the
 *                        file named by RCW_STUB_LOG, waits for all ranks
 *   ncclAllGather        waits for the stream, copies the send buffer device -> shared memory slot `rank`, barrier,
 *                        copies all slots shared memory -> the receive buffer, barrier.  Blocking, not stream-ordered
 *                        beyond that — semantics, not performance.
 * The HIP runtime is NOT linked: its symbols are looked up in the process (the engine has loaded it).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm* ncclComm_t;
typedef int ncclDataType_t;   /* ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7, ncclFloat64 8, ncclBfloat16 9 */

#define STUB_DATA_BYTES (192u << 20)
struct stub_shared {
    volatile int32_t arrived;       /* ranks that have joined */
    volatile int32_t barrier_count;
    volatile int32_t barrier_sense;
    int32_t world;
    char pad[4080];
    unsigned char data[];
};
struct stub_comm {
    int rank, world, fd;
    int local_sense;
    struct stub_shared* sh;
    char name[64];
};

static int (*p_hipMemcpy)(void*, const void*, size_t, int);
static int (*p_hipStreamSynchronize)(void*);
static int hip_ready(void)
{
    if (!p_hipMemcpy) p_hipMemcpy = (int (*)(void*, const void*, size_t, int))dlsym(RTLD_DEFAULT, "hipMemcpy");
    if (!p_hipStreamSynchronize) p_hipStreamSynchronize = (int (*)(void*))dlsym(RTLD_DEFAULT, "hipStreamSynchronize");
    return p_hipMemcpy && p_hipStreamSynchronize;
}

static void stub_log(const char* fmt, ...)
{
    const char* path = getenv("RCW_STUB_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    __builtin_va_list ap;
    __builtin_va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    __builtin_va_end(ap);
    fclose(f);
}

static void barrier(struct stub_comm* c)
{
    c->local_sense = !c->local_sense;
    if (__atomic_add_fetch(&c->sh->barrier_count, 1, __ATOMIC_ACQ_REL) == c->world) {
        __atomic_store_n(&c->sh->barrier_count, 0, __ATOMIC_RELEASE);
        __atomic_store_n(&c->sh->barrier_sense, c->local_sense, __ATOMIC_RELEASE);
    } else {
        for (long spins = 0; __atomic_load_n(&c->sh->barrier_sense, __ATOMIC_ACQUIRE) != c->local_sense; ++spins) {
            usleep(50);
            if (spins > 1200000) { fprintf(stderr, "stub_rccl: barrier timed out (rank %d)\n", c->rank); _exit(97); }
        }
    }
}

__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int* v) { if (v) *v = 0; return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (stub)" : "stub_rccl error"; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static unsigned counter = 0;
    if (!id) return ncclInvalidArgument;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof *id);
    uint64_t a = (uint64_t)getpid() * 0x9e3779b97f4a7c15ull ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)ts.tv_sec << 32) ^ (++counter);
    for (int k = 0; k < 128; ++k) { a ^= a << 13; a ^= a >> 7; a ^= a << 17; id->internal[k] = (char)(a & 0xff); }
    stub_log("unique_id pid=%d first=%02x%02x%02x%02x\n", (int)getpid(), (unsigned char)id->internal[0], (unsigned char)id->internal[1],
             (unsigned char)id->internal[2], (unsigned char)id->internal[3]);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct stub_comm* c = (struct stub_comm*)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank; c->world = nranks;
    snprintf(c->name, sizeof c->name, "/rcw_stub_%02x%02x%02x%02x%02x%02x%02x%02x", (unsigned char)id.internal[0], (unsigned char)id.internal[1],
             (unsigned char)id.internal[2], (unsigned char)id.internal[3], (unsigned char)id.internal[4], (unsigned char)id.internal[5],
             (unsigned char)id.internal[6], (unsigned char)id.internal[7]);
    const size_t bytes = sizeof(struct stub_shared) + STUB_DATA_BYTES;
    if (rank == 0) {
        c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (c->fd < 0 || ftruncate(c->fd, (off_t)bytes) != 0) { free(c); return ncclSystemError; }
    } else {
        for (int tries = 0; (c->fd = shm_open(c->name, O_RDWR, 0600)) < 0; ++tries) {
            usleep(1000);
            if (tries > 60000) { free(c); return ncclSystemError; }
        }
        struct stat st;
        for (int tries = 0; fstat(c->fd, &st) == 0 && (size_t)st.st_size < bytes; ++tries) { usleep(1000); if (tries > 60000) { free(c); return ncclSystemError; } }
    }
    c->sh = (struct stub_shared*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->sh == MAP_FAILED) { free(c); return ncclSystemError; }
    if (rank == 0) c->sh->world = nranks;
    __atomic_add_fetch(&c->sh->arrived, 1, __ATOMIC_ACQ_REL);
    for (long spins = 0; __atomic_load_n(&c->sh->arrived, __ATOMIC_ACQUIRE) < nranks; ++spins) {
        usleep(100);
        if (spins > 600000) { fprintf(stderr, "stub_rccl: rank %d waited for %d ranks in vain\n", rank, nranks); return ncclSystemError; }
    }
    stub_log("comm_init rank=%d world=%d uid=%02x%02x%02x%02x pid=%d\n", rank, nranks, (unsigned char)id.internal[0], (unsigned char)id.internal[1],
             (unsigned char)id.internal[2], (unsigned char)id.internal[3], (int)getpid());
    *comm = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    barrier(c);
    munmap((void*)c->sh, sizeof(struct stub_shared) + STUB_DATA_BYTES);
    close(c->fd);
    if (c->rank == 0) shm_unlink(c->name);
    stub_log("comm_destroy rank=%d\n", c->rank);
    free(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                                                                  ncclComm_t c, void* stream)
{
    static const size_t width[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    if (!c || !sendbuff || !recvbuff || datatype < 0 || datatype > 9) return ncclInvalidArgument;
    const size_t bytes = sendcount * width[datatype];
    if (bytes * (size_t)c->world > STUB_DATA_BYTES) return ncclInvalidArgument;
    if (!hip_ready()) return ncclInternalError;
    if (p_hipStreamSynchronize(stream) != 0) return ncclUnhandledCudaError;                       /* the producer of sendbuff */
    if (p_hipMemcpy((void*)(c->sh->data + (size_t)c->rank * bytes), sendbuff, bytes, 2 /* D2H */) != 0) return ncclUnhandledCudaError;
    barrier(c);
    if (p_hipMemcpy(recvbuff, (const void*)c->sh->data, bytes * (size_t)c->world, 1 /* H2D */) != 0) return ncclUnhandledCudaError;
    barrier(c);                                                                                   /* nobody overwrites a slot another rank still reads */
    stub_log("all_gather rank=%d bytes=%zu\n", c->rank, bytes);
    return ncclSuccess;
}
