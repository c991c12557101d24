"""GGUF v2/v3 reader and writer for the path's loader side.

`Parser` mirrors NFAI.GGUF/Parser.cs: header (:130-143), metadata KV with the 13 value types
(:145-178, :180-242), tensor table (:244-307), data offsets = align(tensor_start + offset) with
`general.alignment` (default 32, :31-32, :47-59, :125-128), then offers the tensors to the model
factories in order (:36-42).  Differences, all on purpose:
  * tensors are zero-copy views over a memory map in their ON-DISK encoding — fp16 is not widened
    to fp32 (AbstractComputeCollection.cs:62-77 does; the HIP kernels widen in-register);
  * Q4_K and Q6_K tensors are returned as raw block bytes (`QuantTensor`) instead of
    `throw new Exception("Unsupported data type")` (Parser.cs:111-114);
  * shapes are returned as (rows, cols) = (ne1, ne0); the reference keeps GGUF order [ne0, ne1].
`GGUFWriter` exists so tests and tools can build model files offline (no checkpoints here).
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass

import numpy as np

from .llama_model import QuantTensor

GGUF_MAGIC = b"GGUF"
# metadata value types (Parser.cs:156-172)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f", T_BOOL: "<?",
           T_U64: "<Q", T_I64: "<q", T_F64: "<d"}
# ggml tensor types (Parser.cs:262-293) -> (block elements, block bytes)
GGML_F32, GGML_F16, GGML_Q4_K, GGML_Q6_K = 0, 1, 12, 14
GGML_BLOCK = {0: (1, 4), 1: (1, 2), 2: (32, 18), 3: (32, 20), 6: (32, 22), 7: (32, 24), 8: (32, 34), 9: (32, 36),
              10: (256, 84), 11: (256, 110), 12: (256, 144), 13: (256, 176), 14: (256, 210), 15: (256, 292),
              24: (1, 1), 25: (1, 2), 26: (1, 4), 27: (1, 8), 28: (1, 8)}
GGML_NAME = {0: "float32", 1: "float16", 2: "Q4_0", 3: "Q4_1", 6: "Q5_0", 7: "Q5_1", 8: "Q8_0", 9: "Q8_1", 10: "Q2_K",
             11: "Q3_K", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K", 15: "Q8_K", 24: "int8", 25: "int16", 26: "int32",
             27: "int64", 28: "float64"}


@dataclass
class TensorInfo:
    name: str
    shape: tuple       # GGUF order (ne0, ne1, ...)
    ggml_type: int
    offset: int        # relative to the start of the tensor data region
    data_offset: int = 0   # absolute file offset

    @property
    def n_elements(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def n_bytes(self) -> int:
        be, bb = GGML_BLOCK[self.ggml_type]
        return self.n_elements // be * bb


class _Reader:
    def __init__(self, buf):
        self.b, self.p = buf, 0

    def take(self, fmt):
        v = struct.unpack_from(fmt, self.b, self.p)[0]
        self.p += struct.calcsize(fmt)
        return v

    def string(self):
        n = self.take("<Q")
        s = bytes(self.b[self.p:self.p + n])
        self.p += n
        return s.decode("utf-8", "replace")

    def value(self, t):
        if t in _SCALAR:
            return self.take(_SCALAR[t])
        if t == T_STR:
            return self.string()
        if t == T_ARR:
            et, n = self.take("<I"), self.take("<Q")
            return [self.value(et) for _ in range(n)]
        raise ValueError("Unsupported metadata value type")  # Parser.cs:173


def align_offset(offset: int, alignment: int) -> int:
    return offset + (alignment - offset % alignment) % alignment  # Parser.cs:125-128


class Parser:
    """≙ NFAI.GGUF.Parser.  `Parse(options)` returns the first provider a factory accepts
    (Parser.cs:22-45); `Read(path)` returns (metadata, tensors) without creating a model."""

    def __init__(self, modelFactories=()):
        self.modelFactories = list(modelFactories)
        self.metadata: dict = {}
        self.tensorInfo: list[TensorInfo] = []
        self.alignment = 32
        self._mm = None

    def Read(self, path: str):
        f = open(path, "rb")
        self._mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        r = _Reader(self._mm)
        if bytes(self._mm[0:4]) != GGUF_MAGIC:
            raise ValueError("Invalid GGUF file format")  # Parser.cs:134-135
        r.p = 4
        self.version = r.take("<I")
        n_tensors = r.take("<Q")
        n_kv = r.take("<Q")
        for _ in range(n_kv):
            key = r.string()
            self.metadata[key] = r.value(r.take("<I"))
        for _ in range(n_tensors):
            name = r.string()
            nd = r.take("<I")
            shape = tuple(r.take("<Q") for _ in range(nd))
            ty = r.take("<I")
            self.tensorInfo.append(TensorInfo(name, shape, ty, r.take("<Q")))
        a = self.metadata.get("general.alignment")
        if isinstance(a, int) and a > 0:
            self.alignment = a
        start = align_offset(r.p, self.alignment)
        tensors = {}
        for ti in self.tensorInfo:
            ti.data_offset = align_offset(ti.offset + start, self.alignment)
            tensors[ti.name] = self._view(ti)
        return self.metadata, tensors

    def _view(self, ti: TensorInfo):
        if ti.ggml_type not in GGML_BLOCK:
            raise ValueError(f"Unsupported data type {ti.ggml_type} for tensor {ti.name}")
        rows = ti.n_elements // ti.shape[0] if len(ti.shape) > 1 else 1
        cols = ti.shape[0]
        raw = np.frombuffer(self._mm, np.uint8, ti.n_bytes, ti.data_offset)
        if ti.ggml_type == GGML_F32:
            a = raw.view(np.float32)
            return a.reshape(rows, cols) if len(ti.shape) > 1 else a
        if ti.ggml_type == GGML_F16:
            a = raw.view(np.float16)
            return a.reshape(rows, cols) if len(ti.shape) > 1 else a
        if ti.ggml_type in (GGML_Q4_K, GGML_Q6_K):
            return QuantTensor(raw, ti.ggml_type, (rows, cols))
        raise ValueError(f"Unsupported data type {GGML_NAME.get(ti.ggml_type, ti.ggml_type)} for tensor {ti.name} "
                         "(kernels exist for F32, F16, Q4_K, Q6_K)")

    def Parse(self, modelOptions, **kw):
        import os
        if not os.path.exists(modelOptions.GGUFPath):
            raise FileNotFoundError(f"File not found: {modelOptions.GGUFPath}")
        metadata, tensors = self.Read(modelOptions.GGUFPath)
        for factory in self.modelFactories:
            ok, model = factory.TryCreate(metadata, tensors, modelOptions, **kw)
            if ok and model is not None:
                return model
        raise RuntimeError("No suitable model factory found for the GGUF file.")

    def GetTensorNames(self):
        return [t.name for t in self.tensorInfo]

    def GetTensorInfo(self):
        return self.tensorInfo


class GGUFWriter:
    def __init__(self, alignment: int = 32, version: int = 3):
        self.kv: list[tuple[str, int, object]] = []
        self.tensors: list[tuple[str, tuple, int, bytes]] = []
        self.alignment, self.version = alignment, version

    def add(self, key: str, value, vtype: int | None = None):
        if vtype is None:
            if isinstance(value, bool):
                vtype = T_BOOL
            elif isinstance(value, int):
                vtype = T_U32
            elif isinstance(value, float):
                vtype = T_F32
            elif isinstance(value, str):
                vtype = T_STR
            elif isinstance(value, (list, tuple)):
                vtype = T_ARR
            else:
                raise TypeError(type(value))
        self.kv.append((key, vtype, value))

    def add_tensor(self, name: str, data, ggml_type: int | None = None, shape: tuple | None = None):
        """data: float32/float16 ndarray [rows][cols] (or 1-D), or raw uint8 block bytes with
        ggml_type + logical (rows, cols)."""
        if isinstance(data, QuantTensor):
            data, ggml_type, shape = data.data, data.ggml_type, data.shape
        a = np.ascontiguousarray(data)
        if ggml_type is None:
            ggml_type = {np.dtype(np.float32): GGML_F32, np.dtype(np.float16): GGML_F16}[a.dtype]
            shape = a.shape
        ne = tuple(reversed(shape))  # GGUF stores ne0 (contiguous) first
        self.tensors.append((name, ne, ggml_type, a.tobytes()))

    @staticmethod
    def _str(s: str) -> bytes:
        b = s.encode("utf-8")
        return struct.pack("<Q", len(b)) + b

    def _val(self, t: int, v) -> bytes:
        if t in _SCALAR:
            return struct.pack(_SCALAR[t], v)
        if t == T_STR:
            return self._str(v)
        if t == T_ARR:
            if not v:
                et = T_U32
            elif isinstance(v[0], str):
                et = T_STR
            elif isinstance(v[0], float):
                et = T_F32
            elif isinstance(v[0], bool):
                et = T_BOOL
            else:
                et = T_I32
            return struct.pack("<IQ", et, len(v)) + b"".join(self._val(et, x) for x in v)
        raise TypeError(t)

    def write(self, path: str):
        head = GGUF_MAGIC + struct.pack("<IQQ", self.version, len(self.tensors), len(self.kv) + 1)
        body = self._str("general.alignment") + struct.pack("<I", T_U32) + struct.pack("<I", self.alignment)
        for k, t, v in self.kv:
            body += self._str(k) + struct.pack("<I", t) + self._val(t, v)
        infos, off = b"", 0
        offsets = []
        for name, ne, ty, data in self.tensors:
            off = align_offset(off, self.alignment)
            offsets.append(off)
            infos += self._str(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne)
            infos += struct.pack("<IQ", ty, off)
            off += len(data)
        with open(path, "wb") as f:
            f.write(head + body + infos)
            start = align_offset(f.tell(), self.alignment)
            f.write(b"\0" * (start - f.tell()))
            for (name, ne, ty, data), o in zip(self.tensors, offsets):
                f.write(b"\0" * (start + o - f.tell()))
                f.write(data)


def write_model(path: str, metadata: dict, tensors: dict, alignment: int = 32):
    """Convenience: metadata dict + {name: ndarray | QuantTensor} -> GGUF file."""
    w = GGUFWriter(alignment)
    for k, v in metadata.items():
        if k == "general.alignment":
            continue
        w.add(k, v)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write(path)
