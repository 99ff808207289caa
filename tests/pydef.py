"""Independent big-integer / hashlib DEFINITIONS of the hot path, used to pin the C oracle.

Nothing here follows the reference's kernel structure: transforms are evaluated from their
mathematical definition, hashes come from hashlib.  Pure-Python loops,
so only for small sizes (k = 512).
"""
import hashlib

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ROOT1 = pow(7, (P - 1) >> 28, P)                      # src/bn254.cpp:36-37
ROOT2 = pow(ROOT1, (1 << 61) - 1, P)                  # src/bn254.cpp:38-39
ROOT2_DEC = 2037444462055058054189478067370099086220733342011840546702672064072905551290


def omegas(k):
    """src/bn254.cpp:51-64"""
    top = 1 << 28
    return pow(ROOT1, top // k, P), pow(ROOT1, top // (2 * k), P), pow(ROOT2, top // (4 * k), P)


def dft(x, w):
    """out[j] = sum_i x[i] w^(ij)  (natural order in and out)"""
    n = len(x)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * w % P
    return [sum(x[i] * pw[(i * j) % n] for i in range(n)) % P for j in range(n)]


def idft(x, w):
    n = len(x)
    ninv = pow(n, -1, P)
    return [v * ninv % P for v in dft(x, pow(w, -1, P))]


def encode(msg, k, n, two_k=False):
    """codeword[j] = Poly(w_4k^j), Poly interpolating msg on the w_k (or w_2k) domain"""
    wk, w2k, w4k = omegas(k)
    coeffs = idft(msg, w2k if two_k else wk)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * w4k % P
    return [sum(c * pw[(i * j) % n] for i, c in enumerate(coeffs)) % P for j in range(n)]


def decode(cw, k, n):
    """c = INTT_n(cw); c[i] += c[i+k] (i<k); buf[0..k) = NTT_k(c[0..k)); buf[k..n) = c[k..n)"""
    wk, _, w4k = omegas(k)
    c = idft(cw, w4k)
    folded = [(c[i] + c[i + k]) % P for i in range(k)]
    return dft(folded, wk) + c[k:]


def leaf(column_elems):
    """message = concat over rows of BE32(limb0)..BE32(limb7); leaf = digest with each 4-byte word reversed"""
    h = hashlib.sha256()
    for v in column_elems:
        for i in range(8):
            h.update(((v >> (32 * i)) & 0xFFFFFFFF).to_bytes(4, "big"))
    d = h.digest()
    return b"".join(d[4 * i:4 * i + 4][::-1] for i in range(8))


def merkle_nodes(leaves):
    """heap layout, missing leaves = 32 zero bytes (include/zkp/merkle_tree.hpp:344-375)"""
    n = len(leaves)
    Pn = 1
    while Pn < n:
        Pn <<= 1
    nodes = [b"\0" * 32] * (2 * Pn - 1)
    for i, l in enumerate(leaves):
        nodes[Pn - 1 + i] = bytes(l)
    for i in range(Pn - 2, -1, -1):
        nodes[i] = hashlib.sha256(nodes[2 * i + 1] + nodes[2 * i + 2]).digest()
    return nodes


def sibling_positions(leaf_indices, total):
    """include/zkp/proof_serializer.hpp:82-117"""
    pos, known = [], set(leaf_indices)
    start, end = total // 2, total
    while start > 0:
        upper = set()
        for i in range(start, end, 2):
            ll = i - start
            kl, kr = ll in known, (ll + 1) in known
            if kl and kr:
                upper.add(ll // 2)
            elif kr:
                pos.append(i); upper.add(ll // 2)
            elif kl:
                pos.append(i + 1); upper.add(ll // 2)
        known = upper
        start, end = (start - 1) // 2, (end - 1) // 2
    return pos


def field_from_keystream(ks32):
    """include/zkp/finite_field_gmp.hpp:66-78"""
    v = int.from_bytes(ks32, "little") >> 2
    return v - P if v >= P else v


class HashRandomEngine:
    """include/zkp/random.hpp:87-146"""

    def __init__(self, seed):
        self.seed, self.state, self.buf, self.off = bytes(seed), 0, b"", -1

    def __call__(self):
        if self.off < 0:
            pre = self.seed if self.state else b""
            self.buf = hashlib.sha256(pre + self.state.to_bytes(8, "little")).digest()
            self.state += 1
            self.off = 31
        b = self.buf[self.off]
        self.off -= 1
        return b


def boost_uniform(eng, rng):
    """SURVEY.md A.7 restatement of boost::random::detail::generate_uniform_int, brange = 255"""
    if rng == 0:
        return 0
    if rng == 255:
        return eng()
    if rng < 255:
        bucket = 256 // (rng + 1)
        while True:
            r = eng() // bucket
            if r <= rng:
                return r
    while True:
        limit = (rng + 1) // 256
        result, mult = 0, 1
        exact = False
        while mult <= limit:
            result += eng() * mult
            if mult * 255 == rng - mult + 1:
                exact = True
                break
            mult *= 256
        if exact:
            return result
        inc = boost_uniform(eng, rng // mult)
        if inc > ((1 << 64) - 1) // mult:
            continue
        inc *= mult
        result = (result + inc) & ((1 << 64) - 1)
        if result < inc or result > rng:
            continue
        return result


def sample_indices(seed, n, t):
    """include/util/portable_sample.hpp:15-33 + sort (src/webgpu_prover.cpp:343-351)"""
    eng = HashRandomEngine(seed)
    a = list(range(n))
    out = []
    for i in range(t):
        j = i + boost_uniform(eng, n - 1 - i)
        a[i], a[j] = a[j], a[i]
        out.append(a[i])
    return sorted(out)


# ---- minimal protobuf reader (wire format only) for the proof envelope ----
def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s; s += 7
        if not c & 0x80:
            return v, i


def pb_fields(b):
    """-> list of (field, wiretype, value) in order of appearance"""
    out, i = [], 0
    while i < len(b):
        tag, i = _varint(b, i)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 2:
            ln, i = _varint(b, i)
            v = bytes(b[i:i + ln]); i += ln
        elif wt == 5:
            v = bytes(b[i:i + 4]); i += 4
        elif wt == 1:
            v = bytes(b[i:i + 8]); i += 8
        else:
            raise ValueError("wire type %d" % wt)
        out.append((f, wt, v))
    return out
