#!/usr/bin/env python
"""Positive probe of the semaphore half of the zero-copy form (csky_external_frame_import_semaphore_fd / _signal) without Vulkan: a DRM
sync object created straight on the render node stands in for the engine's VkSemaphore (an opaque-fd binary semaphore IS a drm_syncobj fd on
amdgpu).  The library imports the fd, signals it behind a march on the march's stream, and the 'engine' side waits on the syncobj through the
kernel (DRM_IOCTL_SYNCOBJ_WAIT): before the signal the wait must time out, after it must succeed."""
import ctypes as C
import fcntl
import glob
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _iowr(nr, size):
    return (3 << 30) | (size << 16) | (ord("d") << 8) | nr


SYNCOBJ_CREATE, SYNCOBJ_DESTROY, SYNCOBJ_HANDLE_TO_FD, SYNCOBJ_WAIT = _iowr(0xBF, 8), _iowr(0xC0, 8), _iowr(0xC1, 16), _iowr(0xC3, 32)
WAIT_FOR_SUBMIT = 2


class SyncObj:
    def __init__(self, node):
        self.drm = os.open(node, os.O_RDWR | os.O_CLOEXEC)
        buf = bytearray(struct.pack("II", 0, 0))
        fcntl.ioctl(self.drm, SYNCOBJ_CREATE, buf)
        self.handle = struct.unpack("II", buf)[0]

    def export_fd(self):
        buf = bytearray(struct.pack("IIiI", self.handle, 0, -1, 0))
        fcntl.ioctl(self.drm, SYNCOBJ_HANDLE_TO_FD, buf)
        return struct.unpack("IIiI", buf)[2]

    def wait(self, timeout_s):
        """True = signalled, False = timed out."""
        arr = (C.c_uint32 * 1)(self.handle)
        deadline = time.clock_gettime_ns(time.CLOCK_MONOTONIC) + int(timeout_s * 1e9)
        buf = bytearray(struct.pack("QqIIII", C.addressof(arr), deadline, 1, WAIT_FOR_SUBMIT, 0, 0))
        try:
            fcntl.ioctl(self.drm, SYNCOBJ_WAIT, buf)
            return True
        except OSError as e:
            import errno
            if e.errno in (errno.ETIME, errno.ETIMEDOUT, errno.EBUSY):
                return False
            raise

    def close(self):
        try:
            fcntl.ioctl(self.drm, SYNCOBJ_DESTROY, bytearray(struct.pack("II", self.handle, 0)))
        finally:
            os.close(self.drm)


def main():
    nodes = sorted(glob.glob("/dev/dri/renderD*"))
    print("render nodes:", nodes)
    if not nodes:
        return 3
    import gvcd_amd
    import ext_frame_roundtrip as X
    from bench import default_params
    W, H = 2048, 1024
    large, small, weather = gvcd_amd.assets.load_default_noise()
    ctx = gvcd_amd.Context(0)
    ctx.set_noise(large, small, weather)
    ctx.render_transmittance(256, 64)
    ctx.set_march(128, 6)
    p, sun = default_params(W, H, (1, 1, 0))
    ctx.render_sky_lut(sun, 200, 100)
    hip = X.load_hip()
    ex = X.ExportedAllocation(hip, 0, W * H * 8)
    L = gvcd_amd.lib()
    ef, dptr = C.c_void_p(), C.c_void_p()
    assert L.csky_external_frame_import_fd(ctx._h, os.dup(ex.fd), C.c_size_t(ex.size), C.c_size_t(0), C.c_size_t(W * H * 8), C.byref(ef), C.byref(dptr)) == 0
    rc = 1
    for node in nodes:
        try:
            so = SyncObj(node)
        except OSError as e:
            print(node, "cannot create a sync object:", e)
            continue
        try:
            fd = so.export_fd()
            r = L.csky_external_frame_import_semaphore_fd(ctx._h, ef, fd)
            if r != 0:
                print(node, "semaphore import refused by the runtime:", L.csky_last_error(ctx._h).decode())
                os.close(fd)
                # the host-side ordering that works on this runtime: a fence behind the march, polled
                import torch
                st = torch.cuda.Stream()
                for _ in range(20):                                        # ~40 ms of marching in front of the fence
                    ctx.render_clouds_device(p, W, (H, 0, 1, 1), dptr.value, W * 8, st.cuda_stream)
                assert L.csky_external_frame_fence(ctx._h, ef, C.c_void_p(st.cuda_stream)) == 0
                early = L.csky_external_frame_ready(ctx._h, ef)
                t0 = time.perf_counter()
                assert L.csky_external_frame_wait(ctx._h, ef) == 0
                dt = time.perf_counter() - t0
                late = L.csky_external_frame_ready(ctx._h, ef)
                print("host-side fence instead: ready right after enqueueing 20 marches: %d; after csky_external_frame_wait (%.1f ms): %d" % (early, dt * 1e3, late))
                if early == 0 and late == 1:
                    rc = 0
                continue
            before = so.wait(0.05)
            import torch
            st = torch.cuda.Stream()
            for _ in range(20):                                            # ~40 ms of marching in front of the signal
                ctx.render_clouds_device(p, W, (H, 0, 1, 1), dptr.value, W * 8, st.cuda_stream)
            r = L.csky_external_frame_signal(ctx._h, ef, C.c_void_p(st.cuda_stream))
            early = so.wait(0.0)                                           # the marches are still running: not yet
            t0 = time.perf_counter()
            after = so.wait(5.0)
            dt = time.perf_counter() - t0
            st.synchronize()
            print("%s: signalled before the signal call: %s; right after enqueueing (marches in flight): %s; after waiting: %s (%.1f ms); signal rc=%d %s"
                  % (node, before, early, after, dt * 1e3, r, L.csky_last_error(ctx._h).decode() if r else ""))
            if not before and after and r == 0:
                rc = 0
        finally:
            so.close()
    L.csky_external_frame_release(ef)
    ex.close()
    ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
