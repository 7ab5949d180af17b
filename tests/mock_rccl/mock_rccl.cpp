// mock_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for the fifteen RCCL entry points libweldacs.so calls
// (csrc/host_comm.inc), so that the multi-rank paths of wa_comm_* can be run with world > 1 on a box that has ONE GPU -- RCCL itself
// refuses two ranks on one device.  LD_PRELOADed in front of librccl by tests/test_gpu_mock_ranks.py; ranks are processes or threads
// that share the GPU and exchange through files in a directory named by the unique id (MOCK_RCCL_DIR, default /tmp/mock_rccl_<uid>).
// Every call is synchronous: wait for the stream, copy to the host, exchange, copy back.  Every wait is bounded (MOCK_RCCL_TIMEOUT_S,
// default 60) and fails with ncclSystemError instead of hanging.  What it checks is the library's own logic around the collectives --
// keys, counts, offsets, owners -- not RCCL.
//
//   g++ -std=c++14 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/mock_rccl/mock_rccl.cpp -L/opt/rocm/lib -lamdhip64 -o libmock_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

struct ncclComm {
    std::string dir;
    int rank, nranks;
    long coll_seq;
    std::vector<long> sent, received;   // per peer
};

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double timeout_s()
{
    const char *v = getenv("MOCK_RCCL_TIMEOUT_S");
    return v && *v ? atof(v) : 60.0;
}
static std::string base_dir()
{
    const char *v = getenv("MOCK_RCCL_DIR");
    if (v && *v) return v;
    char buf[64];
    snprintf(buf, sizeof buf, "/tmp/mock_rccl_%d", (int)getuid());
    return buf;
}
static size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
static bool put_file(const std::string &path, const void *data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;   // appears whole or not at all
}
static bool get_file(const std::string &path, void *data, size_t bytes)
{
    const double t0 = now_s(), limit = timeout_s();
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (now_s() - t0 > limit) { fprintf(stderr, "[mock_rccl] timed out waiting for %s\n", path.c_str()); return false; }
        usleep(200);
    }
    if ((size_t)st.st_size != bytes) { fprintf(stderr, "[mock_rccl] %s: %zu bytes, expected %zu (ranks disagree on a count)\n", path.c_str(), (size_t)st.st_size, bytes); return false; }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}
// every rank's contribution of collective number seq, in rank order
static bool exchange(ncclComm *c, const void *mine, size_t bytes, std::vector<uint8_t> &all)
{
    char name[64];
    const long seq = c->coll_seq++;
    snprintf(name, sizeof name, "/c%ld.r%d", seq, c->rank);
    if (!put_file(c->dir + name, mine, bytes)) return false;
    all.resize(bytes * (size_t)c->nranks);
    for (int r = 0; r < c->nranks; r++) {
        snprintf(name, sizeof name, "/c%ld.r%d", seq, r);
        if (!get_file(c->dir + name, all.data() + bytes * (size_t)r, bytes)) return false;
    }
    return true;
}
static bool to_host(const void *dev, std::vector<uint8_t> &h, size_t bytes, hipStream_t st)
{
    h.resize(bytes);
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
static bool to_dev(void *dev, const void *h, size_t bytes) { return bytes == 0 || hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice) == hipSuccess; }

template <class T>
static void reduce(T *acc, const T *x, size_t n, ncclRedOp_t op)
{
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case ncclSum: acc[i] = acc[i] + x[i]; break;
        case ncclProd: acc[i] = acc[i] * x[i]; break;
        case ncclMax: acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
        case ncclMin: acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
        default: break;
        }
    }
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    unsigned char rnd[16] = {0};
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(rnd, 1, sizeof rnd, f) != sizeof rnd) memset(rnd, 7, sizeof rnd); fclose(f); }
    char *p = id->internal;
    p += sprintf(p, "mock");
    for (unsigned char b : rnd) p += sprintf(p, "%02x", b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || strncmp(id.internal, "mock", 4) != 0) return ncclInvalidArgument;
    ncclComm *c = new ncclComm();
    const std::string base = base_dir();
    mkdir(base.c_str(), 0700);
    c->dir = base + "/" + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) { delete c; return ncclSystemError; }
    c->rank = rank; c->nranks = nranks; c->coll_seq = 0;
    c->sent.assign((size_t)nranks, 0); c->received.assign((size_t)nranks, 0);
    std::vector<uint8_t> all;
    const int32_t me = rank;
    if (!exchange(c, &me, sizeof me, all)) { delete c; return ncclSystemError; }   // everybody is there
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (mock_rccl)";
    case ncclSystemError: return "mock_rccl: file exchange failed or timed out";
    case ncclInvalidArgument: return "mock_rccl: invalid argument";
    default: return "mock_rccl: error";
    }
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *e)
{
    if (e) *e = ncclSuccess;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    const size_t w = type_bytes(t), bytes = w * count;
    if (!c || !w || (t != ncclUint64 && t != ncclFloat64 && t != ncclFloat32 && t != ncclInt32)) return ncclInvalidArgument;
    std::vector<uint8_t> mine, all;
    if (!to_host(send, mine, bytes, st) || !exchange(c, mine.data(), bytes, all)) return ncclSystemError;
    std::vector<uint8_t> acc(all.begin(), all.begin() + (long)bytes);
    for (int r = 1; r < c->nranks; r++) {
        const uint8_t *x = all.data() + bytes * (size_t)r;
        if (t == ncclUint64) reduce((uint64_t *)acc.data(), (const uint64_t *)x, count, op);
        else if (t == ncclFloat64) reduce((double *)acc.data(), (const double *)x, count, op);
        else if (t == ncclFloat32) reduce((float *)acc.data(), (const float *)x, count, op);
        else reduce((int32_t *)acc.data(), (const int32_t *)x, count, op);
    }
    return to_dev(recv, acc.data(), bytes) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t st)
{
    const size_t bytes = type_bytes(t) * sendcount;
    if (!c || !type_bytes(t)) return ncclInvalidArgument;
    std::vector<uint8_t> mine, all;
    if (!to_host(send, mine, bytes, st) || !exchange(c, mine.data(), bytes, all)) return ncclSystemError;
    return to_dev(recv, all.data(), all.size()) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st)
{
    const size_t bytes = type_bytes(t) * count;
    if (!c || !type_bytes(t) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    char name[64];
    snprintf(name, sizeof name, "/b%ld.r%d", c->coll_seq++, root);
    if (c->rank == root) {
        std::vector<uint8_t> mine;
        if (!to_host(send, mine, bytes, st) || !put_file(c->dir + name, mine.data(), bytes)) return ncclSystemError;
        if (recv != send && !to_dev(recv, mine.data(), bytes)) return ncclSystemError;
        return ncclSuccess;
    }
    std::vector<uint8_t> got(bytes);
    if (hipStreamSynchronize(st) != hipSuccess || !get_file(c->dir + name, got.data(), bytes)) return ncclSystemError;
    return to_dev(recv, got.data(), bytes) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank || !type_bytes(t)) return ncclInvalidArgument;
    std::vector<uint8_t> mine;
    if (!to_host(send, mine, type_bytes(t) * count, st)) return ncclSystemError;
    char name[64];
    snprintf(name, sizeof name, "/p%d_%d.%ld", c->rank, peer, c->sent[(size_t)peer]++);
    return put_file(c->dir + name, mine.data(), mine.size()) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank || !type_bytes(t)) return ncclInvalidArgument;
    std::vector<uint8_t> got(type_bytes(t) * count);
    char name[64];
    snprintf(name, sizeof name, "/p%d_%d.%ld", peer, c->rank, c->received[(size_t)peer]++);
    if (hipStreamSynchronize(st) != hipSuccess || !get_file(c->dir + name, got.data(), got.size())) return ncclSystemError;
    return to_dev(recv, got.data(), got.size()) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->nranks;
    return ncclSuccess;
}
ncclResult_t ncclGetVersion(int *version)
{
    if (version) *version = 0;   // "not RCCL"
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

}  // extern "C"
