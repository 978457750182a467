#!/usr/bin/env python
"""Positive round trip of the zero-copy HIP half (csky_external_frame_*) without Vulkan: memory that THIS library did not allocate, handed
over as a POSIX file descriptor, is imported, marched into, and read back through the allocator's own mapping.

The exporter stands in for the engine's VkDeviceMemory (VK_KHR_external_memory_fd, gdext/unverified/zero_copy_vulkan.c): a physical allocation made
with HIP's virtual-memory API (hipMemCreate, requestedHandleType = POSIX fd), exported with hipMemExportToShareableHandle -- on Linux that
is a dma-buf fd, the same kind of object the amdgpu Vulkan drivers hand out -- and mapped by the exporter at its own address
(hipMemAddressReserve / hipMemMap / hipMemSetAccess).  The library never sees that address: it gets the fd."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class MemLocation(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class MemAllocationProp(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", MemLocation), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", AllocFlags)]


class MemAccessDesc(C.Structure):
    _fields_ = [("location", MemLocation), ("flags", C.c_int)]


class ExportedAllocation:
    """A device allocation owned by 'another API': physical memory + an fd for it + the owner's own mapping."""

    def __init__(self, hip, device, nbytes):
        self.hip = hip
        prop = MemAllocationProp()
        prop.type, prop.requestedHandleType = 1, 1                    # hipMemAllocationTypePinned, hipMemHandleTypePosixFileDescriptor
        prop.location.type, prop.location.id = 1, device              # hipMemLocationTypeDevice
        gran = C.c_size_t(0)
        self._chk(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0), "hipMemGetAllocationGranularity")
        g = max(1, gran.value)
        self.size = (nbytes + g - 1) // g * g
        self.handle = C.c_void_p()
        self._chk(hip.hipMemCreate(C.byref(self.handle), C.c_size_t(self.size), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        fd = C.c_int(-1)
        self._chk(hip.hipMemExportToShareableHandle(C.byref(fd), self.handle, 1, C.c_ulonglong(0)), "hipMemExportToShareableHandle")
        self.fd = fd.value
        self.ptr = C.c_void_p()
        self._chk(hip.hipMemAddressReserve(C.byref(self.ptr), C.c_size_t(self.size), C.c_size_t(0), None, C.c_ulonglong(0)), "hipMemAddressReserve")
        self._chk(hip.hipMemMap(self.ptr, C.c_size_t(self.size), C.c_size_t(0), self.handle, C.c_ulonglong(0)), "hipMemMap")
        acc = MemAccessDesc()
        acc.location.type, acc.location.id, acc.flags = 1, device, 3  # read-write
        self._chk(hip.hipMemSetAccess(self.ptr, C.c_size_t(self.size), C.byref(acc), C.c_size_t(1)), "hipMemSetAccess")

    def _chk(self, rc, what):
        if rc != 0:
            self.hip.hipGetErrorString.restype = C.c_char_p
            raise RuntimeError("%s: %s" % (what, self.hip.hipGetErrorString(rc).decode()))

    def read(self, nbytes, offset=0):
        out = np.empty(nbytes, np.uint8)
        self._chk(self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr.value + offset), C.c_size_t(nbytes), 2), "hipMemcpy")   # DeviceToHost
        return out

    def fill(self, byte):
        self._chk(self.hip.hipMemset(self.ptr, byte, C.c_size_t(self.size)), "hipMemset")
        self._chk(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def close(self):
        self.hip.hipMemUnmap(self.ptr, C.c_size_t(self.size))
        self.hip.hipMemAddressFree(self.ptr, C.c_size_t(self.size))
        self.hip.hipMemRelease(self.handle)


def load_hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
    hip.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
    hip.hipMemRelease.argtypes = [C.c_void_p]
    hip.hipMemExportToShareableHandle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_ulonglong]
    hip.hipMemAddressReserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemCreate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemGetAllocationGranularity.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return hip


def main():
    import gvcd_amd
    W, H, OFFSET = 256, 128, 4096
    large, small, weather = gvcd_amd.assets.load_default_noise()
    ctx = gvcd_amd.Context(0)
    ctx.set_noise(large, small, weather)
    ctx.render_transmittance(256, 64)
    ctx.set_march(64, 4)
    from bench import default_params
    p, sun = default_params(W, H, (1, 1, 0))
    ctx.render_sky_lut(sun, 200, 100)
    ref = ctx.render_clouds(p)                                        # the library's own frame, host form
    hip = load_hip()
    ex = ExportedAllocation(hip, 0, OFFSET + W * H * 8)
    ex.fill(0xAB)
    L = gvcd_amd.lib()
    ef, dptr = C.c_void_p(), C.c_void_p()
    rc = L.csky_external_frame_import_fd(ctx._h, os.dup(ex.fd), C.c_size_t(ex.size), C.c_size_t(OFFSET), C.c_size_t(W * H * 8), C.byref(ef), C.byref(dptr))
    if rc != 0:
        print("import refused: rc=%d %s" % (rc, L.csky_last_error(ctx._h).decode()))
        return 2
    print("imported: exporter's mapping 0x%x, library's mapping 0x%x (allocation %d bytes, frame at +%d)" % (ex.ptr.value, dptr.value, ex.size, OFFSET))
    ctx.render_clouds_device(p, W, (H, 0, 1, 1), dptr.value, W * 8, 0)
    ctx.sync()
    got = ex.read(W * H * 8, OFFSET).view(np.float16).reshape(H, W, 4)
    head = ex.read(OFFSET, 0)
    same = bool((got.view(np.uint16) == ref.view(np.uint16)).all())
    print("frame read through the exporter's mapping == the library's host-form frame: %s; bytes before the frame untouched: %s" % (same, bool((head == 0xAB).all())))
    L.csky_external_frame_release(ef)
    ex.close()
    rc = 0 if same and (head == 0xAB).all() else 1
    if "--time" in sys.argv:
        rc |= time_c3(ctx, hip, L)
    ctx.close()
    return rc


def time_c3(ctx, hip, L, frames=200):
    """The headline loop (C3, two frames in flight) with the two frames living in IMPORTED allocations against the same loop on hipMalloc'ed ones."""
    import time
    import torch
    from bench import default_params
    W, H = 2048, 1024
    p, sun = default_params(W, H, (1, 1, 0))
    ctx.set_march(128, 6)
    ctx.set_frames_in_flight(2)
    streams = [torch.cuda.Stream() for _ in range(2)]
    own = [torch.zeros((H, W, 4), dtype=torch.int16, device="cuda:0") for _ in range(2)]
    exs = [ExportedAllocation(hip, 0, W * H * 8) for _ in range(2)]
    efs = [ctx.import_external_frame(os.dup(ex.fd), ex.size, 0, W * H * 8) for ex in exs]      # (the Python wrapper of the same entry point)
    ptrs = [ef.ptr for ef in efs]

    def loop(targets, n):
        for k in range(n):
            st = streams[k & 1].cuda_stream
            ctx.render_sky_lut_device(sun, 200, 100, st)
            ctx.render_clouds_device(p, W, (H, 0, 1, 1), targets[k & 1], W * 8, st)
        torch.cuda.synchronize()

    res = {}
    for name, targets in (("hipMalloc", [t.data_ptr() for t in own]), ("imported fd", ptrs), ("hipMalloc again", [t.data_ptr() for t in own]), ("imported fd again", ptrs)):
        loop(targets, 10)
        t0 = time.perf_counter()
        loop(targets, frames)
        res[name] = (time.perf_counter() - t0) / frames * 1e3
        print("C3, two frames in flight, frames in %-18s %.3f ms per frame" % (name + ":", res[name]))
    a = np.frombuffer(exs[1].read(W * H * 8), np.uint16)
    same = bool((a == own[1].cpu().numpy().view(np.uint16).reshape(-1)).all())
    print("last frame in the imported allocation == last frame in the hipMalloc one: %s" % same)
    efs[0].fence(streams[0].cuda_stream)
    efs[0].wait()
    assert efs[0].ready()
    for ef in efs:
        ef.release()
    for ex in exs:
        ex.close()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
