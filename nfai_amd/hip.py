"""HipBufferManager + ShaderProperty: the backend surface of NFAI.Vulkan/VulkanBufferManager.cs and
NFAI.Vulkan.Shaders/ShaderProperty.cs, over the C ABI (host arrays are NumPy).

Method names follow the reference (including its spellings `DestoryBuffer`, `BindShaderProprty`)
so code written against the Vulkan classes reads the same against these.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call

_DT = {np.dtype(np.float32): 4, np.dtype(np.uint32): 4, np.dtype(np.float16): 2, np.dtype(np.uint8): 1,
       np.dtype(np.int32): 4}


class HipBufferManager:
    """≙ VulkanBufferManager (VulkanBufferManager.cs:9-509): owns the device context and stream.

    One instance per GPU; not thread-safe (neither is the reference, :474-494).
    """

    def __init__(self, device: int = 0, stream: int | None = None):
        h = _lib.H()
        if stream is None:
            call("nfai_hip_ctx_create", device, C.byref(h))
        else:  # enqueue on a caller-owned hipStream_t (pipeline stages share the RCCL stream)
            call("nfai_hip_ctx_create_on_stream", device, C.c_void_p(stream), C.byref(h))
        self.handle = h
        self.device = device
        info = _lib.DeviceInfo()
        call("nfai_hip_ctx_device_info", self.handle, C.byref(info))
        self.info = info

    # -- VulkanBufferManager.CreateBuffer / DestoryBuffer (:42-88, :90-103)
    def CreateBuffer(self, count: int, dtype=np.float32) -> "DeviceBuffer":
        return DeviceBuffer(self, int(count) * np.dtype(dtype).itemsize)

    def DestoryBuffer(self, buf: "DeviceBuffer") -> None:
        buf.free()

    # -- UploadDeviceConstants (:196-244) / UploadDataToDeviceLocal (:105-125)
    def UploadDeviceConstants(self, buf: "DeviceBuffer", data: np.ndarray, start: int = 0, count: int | None = None) -> None:
        a = np.ascontiguousarray(data)
        n = a.size if count is None else count
        a = a.reshape(-1)[:n]
        call("nfai_hip_buf_upload", self.handle, buf.handle, start * a.itemsize, a.ctypes.data_as(C.c_void_p), a.nbytes)

    UploadDataToDeviceLocal = UploadDeviceConstants

    # -- ReadDeviceBufferData (:283-303)
    def ReadDeviceBufferData(self, buf: "DeviceBuffer", count: int, dtype=np.float32, start: int = 0) -> np.ndarray:
        out = np.empty(int(count), dtype)
        call("nfai_hip_buf_download", self.handle, buf.handle, start * out.itemsize, out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    # -- CopyBuffer (:305-318)
    def CopyBuffer(self, src: "DeviceBuffer", dst: "DeviceBuffer", size_bytes: int) -> None:
        call("nfai_hip_buf_copy", self.handle, dst.handle, 0, src.handle, 0, size_bytes)

    def Synchronize(self) -> None:  # ≙ vkQueueWaitIdle
        call("nfai_hip_ctx_synchronize", self.handle)

    def TimerBegin(self) -> None:
        call("nfai_hip_timer_begin", self.handle)

    def TimerEnd(self) -> float:
        ms = C.c_float()
        call("nfai_hip_timer_end", self.handle, C.byref(ms))
        return ms.value

    def WrapDevicePointer(self, ptr: int, nbytes: int) -> "DeviceBuffer":
        return DeviceBuffer(self, nbytes, wrap_ptr=ptr)

    def UploadWeight(self, ggml_type: int, data: np.ndarray, n_rows: int, n_cols: int) -> "DeviceBuffer":
        a = np.ascontiguousarray(data)
        h = _lib.H()
        call("nfai_hip_weight_upload", self.handle, ggml_type, n_rows, n_cols, a.ctypes.data_as(C.c_void_p), C.byref(h))
        return DeviceBuffer(self, a.nbytes, adopt=h)

    def XcdShares(self):
        """(shares[8], probe_us[8]): how the long streaming launches' rows are dealt to the XCDs on this device (zeros: not measured / off)."""
        import ctypes as C
        sh, us = (C.c_uint16 * 8)(), (C.c_float * 8)()
        call("nfai_hip_ctx_xcd_shares", self.handle, sh, us)
        return list(sh), [round(float(v), 2) for v in us]

    def Dispose(self) -> None:  # (:499-509)
        if self.handle:
            call("nfai_hip_ctx_destroy", self.handle)
            self.handle = None


class DeviceBuffer:
    def __init__(self, mgr: HipBufferManager, nbytes: int, wrap_ptr: int | None = None, adopt=None):
        self.mgr = mgr
        self.nbytes = int(nbytes)
        if adopt is not None:
            self.handle = adopt
        else:
            h = _lib.H()
            if wrap_ptr is None:
                call("nfai_hip_buf_alloc", mgr.handle, self.nbytes, C.byref(h))
            else:
                call("nfai_hip_buf_wrap", mgr.handle, C.c_void_p(wrap_ptr), self.nbytes, C.byref(h))
            self.handle = h

    @property
    def device_ptr(self) -> int:
        p, n = C.c_void_p(), C.c_uint64()
        call("nfai_hip_buf_info", self.mgr.handle, self.handle, C.byref(p), C.byref(n))
        return p.value

    def free(self) -> None:
        if self.handle is not None:
            call("nfai_hip_buf_free", self.mgr.handle, self.handle)
            self.handle = None


class ShaderProperty:
    """≙ ShaderProperty<T> (ShaderProperty.cs:8-263): a typed device buffer handle.

    * `BindShaderProprty(other)` frees this property's own buffer and aliases `other`'s with no
      reference count (ShaderProperty.cs:95-108) — chaining ops without copies.
    * `SetValue` / `GetValue` move whole arrays through the host (ShaderProperty.cs:110-182).
    """

    def __init__(self, mgr: HipBufferManager, count: int = 1, dtype=np.float32, name: str = ""):
        self.mgr = mgr
        self.dtype = np.dtype(dtype)
        self.Count = int(count)
        self.Name = name
        self.buffer = mgr.CreateBuffer(self.Count, self.dtype)
        self._owns = True

    @property
    def handle(self):
        return self.buffer.handle

    def BindShaderProprty(self, other: "ShaderProperty") -> None:
        if self._owns:
            self.mgr.DestoryBuffer(self.buffer)
        self.buffer = other.buffer
        self._owns = False
        self.Count = other.Count
        self.dtype = other.dtype

    def SetValue(self, value, start: int = 0, count: int | None = None) -> None:
        a = np.ascontiguousarray(value, self.dtype)
        self.mgr.UploadDeviceConstants(self.buffer, a, start, count)

    def GetValue(self) -> np.ndarray:
        return self.mgr.ReadDeviceBufferData(self.buffer, self.Count, self.dtype)

    def TransferTo(self, target: "ShaderProperty", start: int = 0) -> None:
        # ShaderProperty.cs:20-30 reads to host and re-uploads; a device copy moves the same bytes
        n = min(self.Count, target.Count)
        call("nfai_hip_buf_copy", self.mgr.handle, target.handle, start * self.dtype.itemsize, self.handle, 0,
             n * self.dtype.itemsize)
