// Per-step trajectory frames of a sampling run, taken off the device while the next reverse step computes
// (SURVEY.md 8f-3).  Reference: Denoiser.write runs after every reverse step and writes every system's Atoms to its ASE
// trajectory from the host (relaxation/diffusers/denoising_torch.py:66-82, 358-367, 469-477; relaxation/ase_utils.py:19-48):
// one blocking .cpu() per step.  Here the sampling loop stays ONE library call (adf_sample_traj / adf_eqv2_sample_traj):
// after a step's update kernel the positions are snapshot into one of two device staging buffers on the COMPUTE stream
// (2.4 MB for 1000 systems: ~1 us), and a second stream copies that buffer into a slot of a pinned host ring; an event per
// slot tells a host writer thread (adsorbdiff_amd/trajectory.py) that a frame has landed.  The compute stream only ever
// waits for the copy that used the same staging buffer two steps earlier; the host thread that enqueues the loop blocks
// only when the ring is full (the writer has not released the slot of frame index - slots yet).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "common.h"

struct adf_frames {
    int device;
    int64_t frame_floats;
    int32_t slots;
    float* stage[2];
    float* ring;                 // pinned host memory [slots][frame_floats]
    hipStream_t copy_stream;
    hipEvent_t staged[2];        // compute stream: snapshot into stage[k] done
    hipEvent_t drained[2];       // copy stream: stage[k] has been read out
    bool drained_live[2];
    std::vector<hipEvent_t>* landed;      // per slot: the device-to-host copy has completed
    std::vector<int64_t>* slot_frame;     // frame index held by a slot, -1 = free
    int64_t pushed;                        // frames pushed so far (push order = 0, 1, 2, ...); written and read under mu
    bool aborted;                          // set by adf_frames_abort (the writer died / the run was cancelled): wakes every waiter
    std::mutex* mu;
    std::condition_variable* cv;
};

extern "C" int32_t adf_frames_create(int32_t device, int64_t frame_floats, int32_t slots, adf_frames_t* out) {
    if (!out || frame_floats <= 0 || slots < 2) { adf_set_error("frames_create: bad argument"); return ADF_EINVAL; }
    ADF_HIP_CHECK(hipSetDevice(device));
    adf_frames* f = new adf_frames();
    f->device = device; f->frame_floats = frame_floats; f->slots = slots; f->pushed = 0; f->aborted = false;
    f->stage[0] = f->stage[1] = nullptr; f->ring = nullptr; f->copy_stream = nullptr;
    f->staged[0] = f->staged[1] = f->drained[0] = f->drained[1] = nullptr;
    f->landed = new std::vector<hipEvent_t>(slots, (hipEvent_t) nullptr);
    f->slot_frame = new std::vector<int64_t>(slots, -1);
    f->mu = new std::mutex(); f->cv = new std::condition_variable();
    f->drained_live[0] = f->drained_live[1] = false;
    const size_t bytes = sizeof(float) * (size_t)frame_floats;
    hipError_t e = hipMalloc(&f->stage[0], bytes);
    if (e == hipSuccess) e = hipMalloc(&f->stage[1], bytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&f->ring), bytes * (size_t)slots, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&f->staged[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->drained[k], hipEventDisableTiming);
    }
    for (int s = 0; s < slots && e == hipSuccess; ++s) e = hipEventCreateWithFlags(&(*f->landed)[s], hipEventDisableTiming);
    if (e != hipSuccess) {
        adf_set_error("frames_create: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        // release whatever was created (null / zero handles are skipped)
        for (int k = 0; k < 2; ++k) {
            if (f->stage[k]) (void)hipFree(f->stage[k]);
            if (f->staged[k]) (void)hipEventDestroy(f->staged[k]);
            if (f->drained[k]) (void)hipEventDestroy(f->drained[k]);
        }
        for (hipEvent_t ev : *f->landed) if (ev) (void)hipEventDestroy(ev);
        if (f->ring) (void)hipHostFree(f->ring);
        if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
        delete f->landed; delete f->slot_frame; delete f->mu; delete f->cv;
        delete f;
        *out = nullptr;
        return e == hipErrorOutOfMemory ? ADF_EOOM : ADF_EHIP;
    }
    *out = f;
    return ADF_OK;
}

extern "C" int32_t adf_frames_destroy(adf_frames_t f) {
    if (!f) return ADF_OK;
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(f->copy_stream);
    for (int k = 0; k < 2; ++k) { (void)hipEventDestroy(f->staged[k]); (void)hipEventDestroy(f->drained[k]); (void)hipFree(f->stage[k]); }
    for (hipEvent_t ev : *f->landed) (void)hipEventDestroy(ev);
    (void)hipHostFree(f->ring);
    (void)hipStreamDestroy(f->copy_stream);
    delete f->landed; delete f->slot_frame; delete f->mu; delete f->cv;
    delete f;
    return ADF_OK;
}

// Enqueue frame number f->pushed (the caller's frames are numbered in push order).  Returns ADF_EINVAL once the ring has been
// aborted (adf_frames_abort: the consumer is gone) instead of waiting for a slot that will never be released.
int32_t adf_frames_push_impl(adf_frames* f, const float* src, hipStream_t s) {
    int64_t index;
    int slot;
    {   // the ring slot must have been released by the writer (blocks only when the writer is a whole ring behind)
        std::unique_lock<std::mutex> lk(*f->mu);
        index = f->pushed;
        slot = (int)(index % f->slots);
        f->cv->wait(lk, [&] { return f->aborted || (*f->slot_frame)[slot] < 0; });
        if (f->aborted) { adf_set_error("frames_push: the ring was aborted (its consumer stopped)"); return ADF_EINVAL; }
        (*f->slot_frame)[slot] = index;
    }
    const int k = (int)(index & 1);
    const size_t bytes = sizeof(float) * (size_t)f->frame_floats;
    hipError_t e = hipSuccess;
    if (f->drained_live[k]) e = hipStreamWaitEvent(s, f->drained[k], 0);   // stage[k]'s previous frame is out
    if (e == hipSuccess) e = hipMemcpyAsync(f->stage[k], src, bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipEventRecord(f->staged[k], s);
    if (e == hipSuccess) e = hipStreamWaitEvent(f->copy_stream, f->staged[k], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(f->ring + (size_t)slot * f->frame_floats, f->stage[k], bytes, hipMemcpyDeviceToHost,
                                            f->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(f->drained[k], f->copy_stream);
    if (e == hipSuccess) { f->drained_live[k] = true; e = hipEventRecord((*f->landed)[slot], f->copy_stream); }
    {
        std::lock_guard<std::mutex> lk(*f->mu);
        if (e == hipSuccess) f->pushed = index + 1;
        else (*f->slot_frame)[slot] = -1;   // the frame was not enqueued: the slot is free again, `pushed` does not advance
    }
    f->cv->notify_all();
    if (e != hipSuccess) {
        adf_set_error("frames_push: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? ADF_EOOM : ADF_EHIP;
    }
    return ADF_OK;
}

extern "C" int32_t adf_frames_push(adf_frames_t f, const float* src, void* stream) {
    if (!f || !src) { adf_set_error("frames_push: null argument"); return ADF_EINVAL; }
    return adf_frames_push_impl(f, src, (hipStream_t)stream);
}

// Host side (the writer thread): block until frame `index` has been pushed AND has landed in the ring, or until
// `timeout_ms` has passed (returns ADF_OK with *host_ptr = NULL: the writer polls its stop flag and calls again).
extern "C" int32_t adf_frames_wait(adf_frames_t f, int64_t index, int32_t timeout_ms, const float** host_ptr) {
    if (!f || !host_ptr || index < 0) { adf_set_error("frames_wait: bad argument"); return ADF_EINVAL; }
    *host_ptr = nullptr;
    const int slot = (int)(index % f->slots);
    {
        std::unique_lock<std::mutex> lk(*f->mu);
        const bool ok = f->cv->wait_for(lk, std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 1),
                                        [&] { return f->aborted || f->pushed > index; });
        if (!ok || f->pushed <= index) return ADF_OK;   // time-out, or aborted before this frame was pushed
        if ((*f->slot_frame)[slot] != index) { adf_set_error("frames_wait: frame %lld is no longer in the ring", (long long)index); return ADF_EINVAL; }
    }
    ADF_HIP_CHECK(hipSetDevice(f->device));
    ADF_HIP_CHECK(hipEventSynchronize((*f->landed)[slot]));
    *host_ptr = f->ring + (size_t)slot * f->frame_floats;
    return ADF_OK;
}

extern "C" int32_t adf_frames_release(adf_frames_t f, int64_t index) {
    if (!f || index < 0) { adf_set_error("frames_release: bad argument"); return ADF_EINVAL; }
    const int slot = (int)(index % f->slots);
    {
        std::lock_guard<std::mutex> lk(*f->mu);
        if ((*f->slot_frame)[slot] == index) (*f->slot_frame)[slot] = -1;
    }
    f->cv->notify_all();
    return ADF_OK;
}

// Cancel the ring: every present and future adf_frames_push fails (ADF_EINVAL) instead of waiting for a free slot, every
// adf_frames_wait returns at once.  Called by the host writer when it stops early (I/O error) and by a cancelled run.
extern "C" int32_t adf_frames_abort(adf_frames_t f) {
    if (!f) return ADF_OK;
    {
        std::lock_guard<std::mutex> lk(*f->mu);
        f->aborted = true;
    }
    f->cv->notify_all();
    return ADF_OK;
}

extern "C" int64_t adf_frames_pushed(adf_frames_t f) {
    if (!f) return -1;
    std::lock_guard<std::mutex> lk(*f->mu);
    return f->pushed;
}
