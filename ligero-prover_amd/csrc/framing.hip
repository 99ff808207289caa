// framing.hip -- proof file framing (gzip level 6).
#include <algorithm>
#include <cstring>
#include "../../include/lig_hip.h"

// =====================================================================================================================
// Proof file framing: gzip level 6 of the envelope (src/webgpu_prover.cpp:437-457, src/webgpu_verifier.cpp:249-253)
#include <zlib.h>
extern "C" {

size_t lig_proof_gzip_bound(size_t len) { return (size_t)compressBound((uLong)len) + 32; }

int lig_proof_gzip(const uint8_t* env, size_t len, uint8_t* out, size_t cap, size_t* out_len) {
    if ((!env && len) || !out || !out_len) return LIG_E_ARG;
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return LIG_E_NOMEM;   // 15 + 16: gzip wrapper
    size_t in_pos = 0, out_pos = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {                       // avail_in / avail_out are 32-bit: feed in slices
        const size_t in_now = std::min(len - in_pos, (size_t)1 << 30), out_now = std::min(cap - out_pos, (size_t)1 << 30);
        z.next_in = const_cast<Bytef*>(env + in_pos); z.avail_in = (uInt)in_now;
        z.next_out = out + out_pos; z.avail_out = (uInt)out_now;
        rc = deflate(&z, in_pos + in_now == len ? Z_FINISH : Z_NO_FLUSH);
        in_pos += in_now - z.avail_in; out_pos += out_now - z.avail_out;
        if (rc == Z_STREAM_ERROR || (rc != Z_STREAM_END && out_pos == cap)) { deflateEnd(&z); return rc == Z_STREAM_ERROR ? LIG_E_ARG : LIG_E_NOMEM; }
    }
    deflateEnd(&z);
    *out_len = out_pos;
    return LIG_OK;
}

size_t lig_proof_gunzip_size(const uint8_t* gz, size_t len) {
    if (!gz || len < 18 || gz[0] != 0x1f || gz[1] != 0x8b) return 0;
    return (size_t)gz[len - 4] | ((size_t)gz[len - 3] << 8) | ((size_t)gz[len - 2] << 16) | ((size_t)gz[len - 1] << 24);
}

int lig_proof_gunzip(const uint8_t* gz, size_t len, uint8_t* out, size_t cap, size_t* out_len) {
    if (!gz || !out || !out_len) return LIG_E_ARG;
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return LIG_E_NOMEM;
    size_t in_pos = 0, out_pos = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        const size_t in_now = std::min(len - in_pos, (size_t)1 << 30), out_now = std::min(cap - out_pos, (size_t)1 << 30);
        z.next_in = const_cast<Bytef*>(gz + in_pos); z.avail_in = (uInt)in_now;
        z.next_out = out + out_pos; z.avail_out = (uInt)out_now;
        rc = inflate(&z, Z_NO_FLUSH);
        in_pos += in_now - z.avail_in; out_pos += out_now - z.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); return rc == Z_BUF_ERROR && out_pos == cap ? LIG_E_NOMEM : LIG_E_ARG; }
        if (rc == Z_OK && in_now == z.avail_in && out_now == z.avail_out) { inflateEnd(&z); return out_pos == cap ? LIG_E_NOMEM : LIG_E_ARG; }   // no progress
    }
    inflateEnd(&z);
    *out_len = out_pos;
    return LIG_OK;
}

}  // extern "C"


