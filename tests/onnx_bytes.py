"""Byte-level encoder of the ONNX subset the weight reader parses (protobuf wire format by hand: no `onnx`
package in the image).  Test infrastructure only."""
import struct

import numpy as np

FLOAT, FLOAT16, BFLOAT16, INT64 = 1, 10, 16, 7


def varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(num: int, wt: int) -> bytes:
    return varint((num << 3) | wt)


def ld(num: int, payload: bytes) -> bytes:      # length-delimited field
    return key(num, 2) + varint(len(payload)) + payload


def vi(num: int, v: int) -> bytes:
    return key(num, 0) + varint(v)


def tensor(name: str, arr: np.ndarray, dtype: int = FLOAT, how: str = "raw", packed_dims: bool = True,
           external=None) -> bytes:
    """TensorProto: dims=1, data_type=2, float_data=4, name=8, raw_data=9, external_data=13, data_location=14.
    how: "raw" | "float_data" | "external" (external = (location, offset, length or None))."""
    dims = arr.shape
    body = b""
    if packed_dims:
        body += ld(1, b"".join(varint(d) for d in dims))
    else:
        body += b"".join(vi(1, d) for d in dims)
    body += vi(2, dtype)
    if how == "raw":
        body += ld(9, encode_values(arr, dtype))
    elif how == "float_data":
        assert dtype == FLOAT
        body += ld(4, np.ascontiguousarray(arr, dtype="<f4").tobytes())
    elif how == "external":
        loc, off, length = external
        for k, v in (("location", loc), ("offset", str(off))) + ((("length", str(length)),) if length is not None else ()):
            body += ld(13, ld(1, k.encode()) + ld(2, v.encode()))
        body += vi(14, 1)
    body += ld(8, name.encode())
    return body


def encode_values(arr: np.ndarray, dtype: int) -> bytes:
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if dtype == FLOAT:
        return a.astype("<f4").tobytes()
    if dtype == FLOAT16:
        return a.astype("<f2").tobytes()
    if dtype == BFLOAT16:
        return (a.view(np.uint32) >> 16).astype("<u2").tobytes()     # values are bf16-exact in the tests
    if dtype == INT64:
        return np.ascontiguousarray(arr, dtype="<i8").tobytes()
    raise ValueError(dtype)


def node(op: str, name: str, inputs, outputs) -> bytes:
    """NodeProto: input=1, output=2, name=3, op_type=4."""
    body = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    return body + ld(3, name.encode()) + ld(4, op.encode())


def model(nodes, initializers) -> bytes:
    """ModelProto{ir_version=1, producer_name=2, graph=7{node=1, name=2, initializer=5}, opset_import=8}."""
    graph = b"".join(ld(1, n) for n in nodes) + ld(2, b"main_graph") + b"".join(ld(5, t) for t in initializers)
    opset = ld(8, ld(1, b"") + vi(2, 17))
    return vi(1, 8) + ld(2, b"cqs-test-encoder") + ld(7, graph) + opset
