/*
 * fr_rccl_plugin.cpp -- the RCCL leg of fr_node (the gather of a row-band sharded frame over xGMI), as its own small
 * shared library: libfractalrenderer_amd_rccl.so, linked against librccl.
 *
 * Why a separate library: librccl.so is 570 MB; a DT_NEEDED entry in libfractalrenderer_amd.so would map and relocate
 * it in every process that renders on ONE GPU (and, next to PyTorch's own bundled copy, twice).  fr_node_create loads
 * this plugin (from the directory libfractalrenderer_amd.so lives in) only when a node of DISTINCT devices asks for
 * the RCCL gather; everything else of the library never touches it.
 *
 * One process, one communicator per device (ncclCommInitAll), one host thread per device: the calls below are made by
 * fr_node's per-device worker threads, each with its own communicator and the stream of its own render context, as
 * RCCL's single-process multi-thread usage prescribes.  Point-to-point only (grouped ncclSend / ncclRecv): the path has
 * ONE exchange step, the gather of disjoint byte ranges to a root, and xGMI is a full mesh -- the n - 1 transfers run on
 * n - 1 separate links.  No reference counterpart: the reference is single-GPU (src/vk_engine.cpp:608).
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stddef.h>
#include <stdio.h>

extern "C" {

/* n communicators, rank k on devices[k]; comms[] receives them as opaque pointers.  0 = ok */
int fr_rccl_init(const int* devices, int n, void** comms, char* err, size_t cap)
{
    ncclComm_t c[64];
    if (n < 1 || n > 64) { snprintf(err, cap, "fr_rccl_init: %d devices", n); return -1; }
    const ncclResult_t r = ncclCommInitAll(c, n, devices);
    if (r != ncclSuccess) { snprintf(err, cap, "ncclCommInitAll failed: %s", ncclGetErrorString(r)); return -1; }
    for (int k = 0; k < n; ++k) comms[k] = (void*)c[k];
    return 0;
}

void fr_rccl_destroy(void** comms, int n)
{
    for (int k = 0; k < n; ++k)
        if (comms[k]) { (void)ncclCommDestroy((ncclComm_t)comms[k]); comms[k] = nullptr; }
}

/* ncclCommAbort: stops the operations a communicator still has enqueued or is blocked in (a receive whose send will never
 * come) and frees it.  fr_node calls it for EVERY communicator of a node when a part of a gather failed after its peers
 * had posted their side, so that no stream is left waiting for a transfer that cannot complete. */
void fr_rccl_abort(void** comms, int n)
{
    for (int k = 0; k < n; ++k)
        if (comms[k]) { (void)ncclCommAbort((ncclComm_t)comms[k]); comms[k] = nullptr; }
}

int fr_rccl_group_start(void) { return ncclGroupStart() == ncclSuccess ? 0 : -1; }

int fr_rccl_group_end(char* err, size_t cap)
{
    const ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) { snprintf(err, cap, "ncclGroupEnd failed: %s", ncclGetErrorString(r)); return -1; }
    return 0;
}

int fr_rccl_send(void* comm, const void* buf, size_t bytes, int peer, void* stream, char* err, size_t cap)
{
    const ncclResult_t r = ncclSend(buf, bytes, ncclUint8, peer, (ncclComm_t)comm, (hipStream_t)stream);
    if (r != ncclSuccess) { snprintf(err, cap, "ncclSend failed: %s", ncclGetErrorString(r)); return -1; }
    return 0;
}

int fr_rccl_recv(void* comm, void* buf, size_t bytes, int peer, void* stream, char* err, size_t cap)
{
    const ncclResult_t r = ncclRecv(buf, bytes, ncclUint8, peer, (ncclComm_t)comm, (hipStream_t)stream);
    if (r != ncclSuccess) { snprintf(err, cap, "ncclRecv failed: %s", ncclGetErrorString(r)); return -1; }
    return 0;
}

int fr_rccl_version(void)
{
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

}  // extern "C"
