"""Batch ("vbn254fr") programs for the prover tests: a tiny assembler producing the op / data blobs that both
lig_synth_job (include/lig_hip.h) and lo_job (oracle/lig_oracle.h) take -- same 32-byte op layout in both."""
import ctypes as C

import numpy as np

OPS = dict(SET=0, SET_SCALAR=1, COPY=2, ADD=3, SUB=4, MUL=5, DIV=6, ADD_CONST=7, SUB_CONST=8, CONST_SUB=9, MUL_CONST=10,
           MONTMUL_CONST=11, ASSERT_EQUAL=12, BIT_DECOMPOSE=13, FREE=14, UPSTREAM_COMPAT=15)
F_WRITE_LIMBS = 1        # lig_batch_op.reserved of SET / SET_SCALAR: written by the write_limbs family (nothing cleared)


class BatchOp(C.Structure):
    _fields_ = [("op", C.c_uint32), ("out", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32), ("len", C.c_uint32),
                ("reserved", C.c_uint32), ("data_off", C.c_uint64)]


class Program:
    def __init__(self):
        self.ops, self.data = [], bytearray()

    def _blob(self, b):
        off = len(self.data)
        self.data += b
        return off

    def _op(self, name, out=0, x=0, y=0, length=0, off=0, flags=0):
        self.ops.append((OPS[name], out, x, y, length, off, flags))

    def upstream_compat(self): self._op("UPSTREAM_COMPAT")       # slices as src/webgpu/buffer_view.cpp:91-95 defines them, from here on

    def set(self, x, values, limbs=False):
        self._op("SET", x=x, length=len(values), off=self._blob(b"".join(int(v).to_bytes(32, "little") for v in values)), flags=F_WRITE_LIMBS if limbs else 0)

    def set_scalar(self, x, v, limbs=False): self._op("SET_SCALAR", x=x, off=self._blob(int(v).to_bytes(32, "little")), flags=F_WRITE_LIMBS if limbs else 0)
    def copy(self, out, x): self._op("COPY", out=out, x=x)
    def add(self, out, x, y): self._op("ADD", out, x, y)
    def sub(self, out, x, y): self._op("SUB", out, x, y)
    def mul(self, out, x, y): self._op("MUL", out, x, y)
    def div(self, out, x, y): self._op("DIV", out, x, y)
    def const(self, name, out, x, c): self._op(name, out, x, off=self._blob(int(c).to_bytes(32, "little")))
    def assert_equal(self, x, y): self._op("ASSERT_EQUAL", x=x, y=y)
    def free(self, x): self._op("FREE", x=x)

    def bit_decompose(self, outs, x):
        self._op("BIT_DECOMPOSE", x=x, length=len(outs), off=self._blob(np.asarray(outs, dtype=np.uint32).tobytes()))

    def pack(self):
        """-> (ops ctypes array, data ctypes array); keep both alive while a job points at them"""
        arr = (BatchOp * max(1, len(self.ops)))()
        for i, (op, out, x, y, length, off, flags) in enumerate(self.ops):
            arr[i].op, arr[i].out, arr[i].x, arr[i].y, arr[i].len, arr[i].data_off, arr[i].reserved = op, out, x, y, length, off, flags
        data = (C.c_uint8 * max(1, len(self.data))).from_buffer_copy(bytes(self.data) or b"\0")
        return arr, data

    def attach(self, job):
        """point a lig_synth_job / lo_job at this program (fields batch_ops, n_batch_ops, batch_data, batch_data_bytes)"""
        arr, data = self.pack()
        job.batch_ops = C.cast(arr, C.c_void_p)
        job.n_batch_ops = len(self.ops)
        job.batch_data = C.cast(data, C.c_void_p)
        job.batch_data_bytes = len(self.data)
        job._keep = (arr, data)
        return job
