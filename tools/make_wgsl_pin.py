#!/usr/bin/env python3
"""INGEST HOOK (round 6): pins the oracle's NTT output order and leaf byte order to the reference's WGSL kernels
EXECUTING (shader/kernels.wgsl.in:58-323, shader/sha256.wgsl:148-228, driven by src/webgpu/engine.cpp:755-882,1514-1686).  No WGSL
compiler / Dawn exists in the build image; on any machine where the reference's `webgpu_context` runs (native Dawn or a browser build):

    ctx.webgpu_init(...); ctx.ntt_init(320, 512, 2048);                       # l = 320, k = 512, n = 2048
    for r in 0..2:   limbs[i] = to_limbs((i*i + 7 + r) mod p) for i in 0..511  # 8 x u32 little-endian per element, host format of device_bignum
                     buf_r = make_device_buffer(n * 32); write_buffer_clear(buf_r, limbs, 512 * 32); encode_ntt_device(buf_r)
    dump  copy_to_host(buf_0) ++ copy_to_host(buf_1) ++ copy_to_host(buf_2)   -> codewords.bin   (3 * 2048 * 32 bytes)
    sha = make sha256_context; sha256_digest_init(sha); sha256_digest_update(sha, buf_r) for r in 0..2; sha256_digest_final(sha, out)
    dump  copy_to_host(out)                                                   -> leaves.bin      (2048 * 32 bytes)

then   python tools/make_wgsl_pin.py codewords.bin leaves.bin > tests/golden/wgsl_encode_k512.json
The tests that read it (tests/test_ref_pins.py::test_encode_and_leaves_equal_the_wgsl_kernels_executing on the CPU oracle,
tests/test_gpu_parity.py::test_hip_encode_and_leaves_equal_the_wgsl_kernels_executing on the HIP path) skip while the file is absent;
DESIGN.md section 5 keeps saying "parity unpinned" for these two orders until then.  This script only packages the dumps (hex + SHA-256)."""
import hashlib
import json
import sys

K, N, ROWS = 512, 2048, 3


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    cw = open(sys.argv[1], "rb").read()
    leaves = open(sys.argv[2], "rb").read()
    if len(cw) != ROWS * N * 32 or len(leaves) != N * 32:
        raise SystemExit("expected %d bytes of codewords and %d bytes of leaves" % (ROWS * N * 32, N * 32))
    out = {"generator": "tools/make_wgsl_pin.py from dumps of the reference's webgpu_context (see the script's docstring)",
           "l": 320, "k": K, "n": N, "rows": ROWS, "message_rule": "row r, element i = (i*i + 7 + r) mod p, i < k",
           "codewords_hex": [cw[r * N * 32:(r + 1) * N * 32].hex() for r in range(ROWS)],
           "codewords_sha256": [hashlib.sha256(cw[r * N * 32:(r + 1) * N * 32]).hexdigest() for r in range(ROWS)],
           "leaves_hex": leaves.hex(), "leaves_sha256": hashlib.sha256(leaves).hexdigest()}
    json.dump(out, sys.stdout)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
