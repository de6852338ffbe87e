"""Julia's default RNG stream (Julia <= 1.6: MersenneTwister = dSFMT-19937), restated so that the reference's seeded experiments can be re-drawn.

TEST INFRASTRUCTURE.  The reference's training scripts draw their experiments from `Random.seed!(1234)` (case2/case2.jl:11) and its README pins
"Julia 1.6" for cases 1-3, Robertson and HyChem; Julia's Random stdlib is not in /root/reference (it ships with Julia), so what follows restates
published algorithms:
  * dSFMT-19937 (Saito & Matsumoto, dSFMT 2.2: `dsfmt_chk_init_by_array`, `do_recursion`, `initial_mask`, `period_certification`; parameter set
    dSFMT-params19937.h), seeded the way `Random.seed!(::MersenneTwister, 1234)` does: `make_seed(1234) = UInt32[1234]` -> init_by_array;
  * `rand(Float64)`: the next double of the close1-open2 stream minus 1 (stdlib/Random/src/RNGs.jl, `rand(r, CloseOpen01())`);
  * `rand!(::MersenneTwister, ::Array{Float32})` (RNGs.jl, the Float16 / Float32 array method): the array's memory is filled with n*4 ÷ 16 * 2
    stream doubles, every 128-bit word gets `u ⊻= u << 26`, the mantissa mask and the exponent of 1.0f0; the < 16-byte tail takes scalar
    `rand(Float32)` = the low 23 bits of the next double's mantissa; then 1 is subtracted;
  * `randn` (stdlib/Random/src/normal.jl): the 256-level ziggurat on the low 52 bits of a stream double (one sign bit, 51-bit magnitude, level =
    low 8 bits of the magnitude), tables as created by randmtzig's `create_ziggurat_tables` (scaled to 51 bits);
    `randn!(::MersenneTwister, ::Array{Float64})` for length >= 13 (Julia >= 1.5, NEWS "#35078") first fills the array with stream doubles and then
    converts them in place, so the unlikely branch's extra draws come AFTER the block; `array_randn=False` gives the element-by-element order
    of Julia <= 1.4.
Pinned below (`self_check`) against the values Julia's own documentation prints for `MersenneTwister(1234)`: `rand(rng, 2)`, `rand!(rng, zeros(5))`
and `randn(rng, ComplexF64)`.
"""
import math

import numpy as np

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1
# dSFMT-params19937.h
N = 191
POS1 = 117
SL1 = 19
MSK1 = 0x000ffafffffffb3f
MSK2 = 0x000ffdfffc90fffd
FIX1 = 0x90014964b32f4329
FIX2 = 0x3b8d12ac548a7c7a
PCV1 = 0x3d84e1ac0dc82880
PCV2 = 0x0000000000000001
SR = 12
LOW_MASK = 0x000FFFFFFFFFFFFF
HIGH_CONST = 0x3FF0000000000000


class DSFMT:
    """The state is N + 1 128-bit words held as 2 (N + 1) uint64 (`u[2 i]`, `u[2 i + 1]`); `block()` advances it by N words = 382 doubles."""

    def __init__(self, key):
        size = (N + 1) * 4
        lag = 11 if size >= 623 else 7 if size >= 68 else 5 if size >= 39 else 3
        mid = (size - lag) // 2
        s = [0x8b8b8b8b] * size

        def f1(x):
            return ((x ^ (x >> 27)) * 1664525) & M32

        def f2(x):
            return ((x ^ (x >> 27)) * 1566083941) & M32

        count = max(len(key) + 1, size)
        r = f1(s[0] ^ s[mid % size] ^ s[(size - 1) % size])
        s[mid % size] = (s[mid % size] + r) & M32
        r = (r + len(key)) & M32
        s[(mid + lag) % size] = (s[(mid + lag) % size] + r) & M32
        s[0] = r
        count -= 1
        i, j = 1, 0
        while j < count and j < len(key):
            r = f1(s[i] ^ s[(i + mid) % size] ^ s[(i + size - 1) % size])
            s[(i + mid) % size] = (s[(i + mid) % size] + r) & M32
            r = (r + key[j] + i) & M32
            s[(i + mid + lag) % size] = (s[(i + mid + lag) % size] + r) & M32
            s[i] = r
            i = (i + 1) % size
            j += 1
        while j < count:
            r = f1(s[i] ^ s[(i + mid) % size] ^ s[(i + size - 1) % size])
            s[(i + mid) % size] = (s[(i + mid) % size] + r) & M32
            r = (r + i) & M32
            s[(i + mid + lag) % size] = (s[(i + mid + lag) % size] + r) & M32
            s[i] = r
            i = (i + 1) % size
            j += 1
        for j in range(size):
            r = f2((s[i] + s[(i + mid) % size] + s[(i + size - 1) % size]) & M32)
            s[(i + mid) % size] ^= r
            r = (r - i) & M32
            s[(i + mid + lag) % size] ^= r
            s[i] = r
            i = (i + 1) % size
        u = [s[2 * k] | (s[2 * k + 1] << 32) for k in range(2 * (N + 1))]          # little-endian 32 -> 64
        for k in range(2 * N):                                                       # initial_mask
            u[k] = (u[k] & LOW_MASK) | HIGH_CONST
        t0, t1 = u[2 * N] ^ FIX1, u[2 * N + 1] ^ FIX2                                # period_certification
        inner = (t0 & PCV1) ^ (t1 & PCV2)
        sh = 32
        while sh > 0:
            inner ^= inner >> sh
            sh >>= 1
        if (inner & 1) != 1:
            u[2 * N + 1] ^= 1                                                        # PCV2 & 1 == 1
        self.u = u

    def block(self):
        """dsfmt_gen_rand_all: returns the 2 N = 382 new 64-bit words (doubles in [1, 2) as bit patterns)."""
        u = self.u
        L0, L1 = u[2 * N], u[2 * N + 1]
        for i in range(N):
            b = i + POS1 if i + POS1 < N else i + POS1 - N
            t0, t1 = u[2 * i], u[2 * i + 1]
            n0 = ((t0 << SL1) & M64) ^ (L1 >> 32) ^ ((L1 << 32) & M64) ^ u[2 * b]
            n1 = ((t1 << SL1) & M64) ^ (L0 >> 32) ^ ((L0 << 32) & M64) ^ u[2 * b + 1]
            L0, L1 = n0, n1
            u[2 * i] = (L0 >> SR) ^ (L0 & MSK1) ^ t0
            u[2 * i + 1] = (L1 >> SR) ^ (L1 & MSK2) ^ t1
        u[2 * N], u[2 * N + 1] = L0, L1
        return u[:2 * N]


def _ziggurat_tables():
    """The 256-level normal ziggurat of stdlib/Random/src/normal.jl (`ki`, `wi`, `fi`; magnitudes scaled to 51 bits): randmtzig's
    `create_ziggurat_tables` recursion x_i = f^-1(v / x_{i+1} + f(x_{i+1})) from `ziggurat_nor_r`, evaluated in extended precision with the exact
    strip area v = r f(r) + int_r^inf f, and rounded once (in double arithmetic with randmtzig's truncated v the tables are 2e-12 off Julia's;
    this way `ki[1]`, `ki[3]` equal the literals 0x0007799ec012f7b2, 0x0006045f4c7de363 and the documented randn values come out bit for bit)."""
    import mpmath as mp
    with mp.workdps(60):
        R = mp.mpf("3.65415288536100879635194725185604664812733315920964488827246397029393565706474")
        f = lambda x: mp.e ** (-x * x / 2)
        v = R * f(R) + mp.sqrt(mp.pi / 2) * mp.erfc(R / mp.sqrt(2))
        NM = mp.mpf(2) ** 51
        ki = [0] * 256
        wi = [0.0] * 256
        fi = [0.0] * 256
        x1 = R
        wi[255] = float(x1 / NM)
        f1 = f(x1)
        fi[255] = float(f1)
        ki[0] = int(mp.floor(x1 * f1 / v * NM))
        wi[0] = float(v / f1 / NM)
        fi[0] = 1.0
        for i in range(254, 0, -1):
            x = mp.sqrt(-2 * mp.log(v / x1 + f1))
            ki[i + 1] = int(mp.floor(x / x1 * NM))
            wi[i] = float(x / NM)
            f1 = f(x)
            fi[i] = float(f1)
            x1 = x
        ki[1] = 0
    return ki, wi, fi


KI, WI, FI = _ziggurat_tables()
NOR_R = 3.6541528853610088
NOR_INV_R = 0.27366123732975828


class MersenneTwister:
    """The consumption order of Julia's MersenneTwister: one linear stream of 52-bit mantissas (the 1002-double cache of RNGs.jl only batches it)."""

    def __init__(self, seed=1234, array_randn=True):
        key = []
        s = int(seed)
        while True:                              # make_seed(n::Integer): base-2^32 digits, least significant first
            key.append(s & M32)
            s >>= 32
            if s == 0:
                break
        self.g = DSFMT(key)
        self.buf = []
        self.pos = 0
        self.array_randn = array_randn
        self.drawn = 0

    def _next_bits(self):
        if self.pos >= len(self.buf):
            self.buf = list(self.g.block())
            self.pos = 0
        v = self.buf[self.pos]
        self.pos += 1
        self.drawn += 1
        return v

    @staticmethod
    def _f64(bits):
        return float(np.array([bits], dtype=np.uint64).view(np.float64)[0])

    def rand(self):
        """rand(Float64): CloseOpen12 - 1."""
        return self._f64(self._next_bits()) - 1.0

    def rand_f64(self, n):
        return np.array([self.rand() for _ in range(n)])

    def rand_f32_array(self, n):
        """rand(Float32, dims) with prod(dims) = n, column-major order."""
        n128 = n * 4 // 16
        words = np.array([self._next_bits() for _ in range(2 * n128)], dtype=np.uint64)
        out = np.empty(n, dtype=np.float32)
        o32 = out.view(np.uint32)
        for i in range(n128):
            u = int(words[2 * i]) | (int(words[2 * i + 1]) << 64)
            u ^= (u << 26) & ((1 << 128) - 1)
            u = (u & 0x007fffff007fffff007fffff007fffff) | 0x3f8000003f8000003f8000003f800000
            for k in range(4):
                o32[4 * i + k] = (u >> (32 * k)) & M32
        for i in range(4 * n128, n):             # scalar rand(Float32) + 1: low 23 mantissa bits of the next double
            o32[i] = (self._next_bits() & 0x007fffff) | 0x3f800000
        return (out - np.float32(1.0)).astype(np.float32)

    def _randn_bits(self, bits):
        ui = bits & LOW_MASK
        rabs = ui >> 1
        idx = rabs & 0xFF
        x = (-rabs if (ui & 1) else rabs) * WI[idx]
        if rabs < KI[idx]:
            return x
        return self._randn_unlikely(idx, rabs, x)

    def _randn_unlikely(self, idx, rabs, x):
        if idx == 0:
            while True:
                xx = -NOR_INV_R * math.log(self.rand())
                yy = -math.log(self.rand())
                if yy + yy > xx * xx:
                    return -NOR_R - xx if ((rabs >> 8) & 1) else NOR_R + xx
        elif (FI[idx - 1] - FI[idx]) * self.rand() + FI[idx] < math.exp(-0.5 * x * x):
            return x
        return self.randn()

    def randn(self):
        return self._randn_bits(self._next_bits())

    def randn_array(self, n):
        """randn(dims...) :: Array{Float64}, column-major."""
        if self.array_randn and n >= 13:
            raw = [self._next_bits() for _ in range(n)]
            return np.array([self._randn_bits(b) for b in raw])
        return np.array([self.randn() for _ in range(n)])

    def randn_f32(self, n):
        """randn(Float32, n): Float32(randn()) element by element."""
        return np.array([self.randn() for _ in range(n)]).astype(np.float32)


def self_check():
    """Known answers from Julia's documentation (Random stdlib docstrings, MersenneTwister(1234))."""
    r = MersenneTwister(1234)
    a = r.rand_f64(2)
    assert a.tolist() == [0.5908446386657102, 0.7667970365022592], a.tolist()
    r = MersenneTwister(1234)
    b = r.rand_f64(5)
    assert b.tolist() == [0.5908446386657102, 0.7667970365022592, 0.5662374165061859, 0.4600853424625171, 0.7940257103317943], b.tolist()
    r = MersenneTwister(1234)
    z = [r.randn(), r.randn()]                   # randn(rng, ComplexF64) = 0.6133070881429037 - 0.6376291670853887im = these / sqrt(2)
    assert z == [0.8673472019512456, -0.9017438158568171], z
    assert KI[0] == 0x0007799ec012f7b2 and KI[2] == 0x0006045f4c7de363, (hex(KI[0]), hex(KI[2]))
    return True


if __name__ == "__main__":
    self_check()
    print("julia_rng: dSFMT-19937 stream, Float64 rand and the ziggurat reproduce the documented MersenneTwister(1234) values")
