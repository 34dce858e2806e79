#!/usr/bin/env python3
"""Writes sycl_blas_amd/csrc/gemm_f32_kloop.hpp: the steady-state K loop of the fp32 256 x 256 x 16 tile kernel
as one block of gfx950 assembly per transpose combination.

Why assembly: in the compiled loop hipcc decides where the LDS reads, LDS writes, global loads and their waits
sit between the matrix instructions, and a scheduling pattern only orders what it claims.  Here every
instruction of a K step has a fixed slot:

  * 64 v_mfma_f32_32x32x2_f32 per wavefront and step (8 accumulator tiles x 8 k pairs), in the order of
    GemmBody::mma_group -- for every C element k ascends, so the result keeps the bits of the compiled loop;
  * the fragments of group g + 1 are requested behind the first six matrix instructions of group g and
    waited for behind its last one (ten matrix instructions of slack);
  * the tile that arrived in registers during the previous step is written to the other LDS buffer behind
    matrix instructions 6.. of group 0, the next tile is requested behind those of group 1;
  * all LDS addresses are a per-lane base register plus an immediate (the two buffers are the two halves of an
    unrolled pair of steps), all global addresses a scalar base plus a per-lane 32-bit offset: no address
    arithmetic on the vector unit inside the loop;
  * one s_barrier per step, in front of the last group, as in the compiled loop.

The LDS images are exactly those of TileLoader / GemmTile in gemm_mfma.hpp, so the prologue, the last steps
and the epilogue stay in C++ around the block.
Run from the repository root:  python tools/gen_f32_kloop.py
"""
import os

BM = BN = 256
BK = 16
NT = 512
S_MN = 256            # row stride (floats) of an [k][mn] image
S_K = BK + 2          # row stride of an [mn][k] image
ELEMS_MN = BK * S_MN  # 4096
ELEMS_K = BM * S_K    # 4608


class Variant:
    def __init__(self, ta, tb, mi, ni):
        self.ta, self.tb, self.mi, self.ni = ta, tb, mi, ni
        self.a_kmajor = ta          # A is K-contiguous when transposed
        self.b_kmajor = not tb      # B is K-contiguous when NOT transposed
        self.a_elems = ELEMS_K if self.a_kmajor else ELEMS_MN
        self.b_elems = ELEMS_K if self.b_kmajor else ELEMS_MN
        self.stage_bytes = (self.a_elems + self.b_elems) * 4
        # register map
        self.acc = lambda i, j: 16 * (i * ni + j)
        self.frag_base = 128                      # two sets of 12
        self.set_regs = 2 * (mi + ni)
        self.st_a = 128 + 2 * self.set_regs       # 152: two vectors of four
        self.st_b = self.st_a + 8                 # 160
        nxt = self.st_b + 8                       # 168: temporaries of the block start here
        self.rd_a, nxt = self.alloc_read(self.a_kmajor, mi, nxt)
        self.rd_b, nxt = self.alloc_read(self.b_kmajor, ni, nxt)
        self.wr_a, nxt = self.alloc_write(self.a_kmajor, nxt)
        self.wr_b, nxt = self.alloc_write(self.b_kmajor, nxt)
        self.first_temp, self.end_temp = self.st_b + 8, nxt

    @staticmethod
    def alloc_read(kmajor, tiles, nxt):
        # K-major image: one base, tiles by immediate.  MN-major: one base per tile (128 B apart, below the
        # 256-byte granule of the ds_read2st64 offsets)
        n = 1 if kmajor else tiles
        return list(range(nxt, nxt + n)), nxt + n

    @staticmethod
    def alloc_write(kmajor, nxt):
        # K-major image: ds_write2_b32 has 8-bit offsets only -> one base per (vector, buffer).  MN-major: one base.
        n = 4 if kmajor else 1
        return list(range(nxt, nxt + n)), nxt + n

    def fa(self, s, i, e):
        return self.frag_base + s * self.set_regs + 2 * i + e

    def fb(self, s, j, e):
        return self.frag_base + s * self.set_regs + 2 * self.mi + 2 * j + e

    # ---- instruction text
    def reads(self, g, s, buf):
        """the six fragment reads of group g into set s from LDS buffer buf"""
        out = []
        for op, kmajor, tiles, rd, dst in (("a", self.a_kmajor, self.mi, self.rd_a, self.fa), ("b", self.b_kmajor, self.ni, self.rd_b, self.fb)):
            for t in range(tiles):
                d = dst(s, t, 0)
                if kmajor:
                    off = t * 32 * S_K * 4 + g * 16 + buf * self.stage_bytes
                    out.append(f"ds_read_b64 v[{d}:{d + 1}], v{rd[0]} offset:{off}")
                else:
                    assert self.stage_bytes % 256 == 0
                    o0 = 16 * g + buf * self.stage_bytes // 256
                    assert o0 + 8 <= 255
                    out.append(f"ds_read2st64_b32 v[{d}:{d + 1}], v{rd[t]} offset0:{o0} offset1:{o0 + 8}")
        return out

    def writes(self, buf):
        out = []
        for kmajor, wr, st in ((self.a_kmajor, self.wr_a, self.st_a), (self.b_kmajor, self.wr_b, self.st_b)):
            for i in range(2):
                r = st + 4 * i
                if kmajor:
                    base = wr[2 * i + buf]
                    out.append(f"ds_write2_b32 v{base}, v{r}, v{r + 2} offset1:1")
                    out.append(f"ds_write2_b32 v{base}, v{r + 1}, v{r + 3} offset0:2 offset1:3")
                else:
                    off = i * 8 * S_MN * 4 + buf * self.stage_bytes
                    out.append(f"ds_write_b128 v{wr[0]}, v[{r}:{r + 3}] offset:{off}")
        return out

    def loads(self):
        out = []
        for st, goff, base in ((self.st_a, "%[goa", "s[44:45]"), (self.st_b, "%[gob", "s[46:47]")):
            for i in range(2):
                out.append(f"global_load_dwordx4 v[{st + 4 * i}:{st + 4 * i + 3}], {goff}{i}], {base}")
        # move both bases one K step on (scalar unit)
        out[-1] += "\ns_add_u32 s44, s44, s48\ns_addc_u32 s45, s45, s49\ns_add_u32 s46, s46, s50\ns_addc_u32 s47, s47, s51"
        return out

    def mfmas(self, s):
        out = []
        for e in range(2):
            for i in range(self.mi):
                for j in range(self.ni):
                    a = self.acc(i, j)
                    out.append(f"v_mfma_f32_32x32x2_f32 v[{a}:{a + 15}], v{self.fb(s, j, e)}, v{self.fa(s, i, e)}, v[{a}:{a + 15}]")
        return out

    def step(self, cur):
        """one K step computing out of buffer `cur`"""
        nxt = cur ^ 1
        lines = []
        side = {0: self.reads(1, 1, cur), 1: self.reads(2, 0, cur), 2: self.reads(3, 1, cur), 3: self.reads(0, 0, nxt)}
        writes = self.writes(nxt)
        loads = self.loads()
        for g in range(4):
            m = self.mfmas(g & 1)
            extra = list(side[g])
            if g == 0:
                extra += ["s_waitcnt vmcnt(0)\n" + writes[0]] + writes[1:]
            if g == 1:
                extra += loads
            for n, ins in enumerate(m):
                lines.append(ins)
                if n < len(extra):
                    lines.append(extra[n])
            assert len(extra) <= len(m)
            if g == 0:
                lines.append(f"s_waitcnt lgkmcnt({len(writes)})")  # the reads of group 1 (issued before the writes) have landed
            elif g == 2:
                lines.append("s_waitcnt lgkmcnt(0)")
                lines.append("s_barrier")
            else:
                lines.append("s_waitcnt lgkmcnt(0)")
        return lines

    def setup(self):
        """derive the per-tile / per-buffer base registers from the four bases the caller passes"""
        out = []
        if self.a_kmajor:
            out.append(f"v_mov_b32 v{self.rd_a[0]}, %[rda]")
        else:
            out += [f"v_add_u32 v{r}, {128 * t}, %[rda]" for t, r in enumerate(self.rd_a)]
        if self.b_kmajor:
            out.append(f"v_mov_b32 v{self.rd_b[0]}, %[rdb]")
        else:
            out += [f"v_add_u32 v{r}, {128 * t}, %[rdb]" for t, r in enumerate(self.rd_b)]
        for kmajor, wr, name in ((self.a_kmajor, self.wr_a, "%[wra]"), (self.b_kmajor, self.wr_b, "%[wrb]")):
            if kmajor:
                for i in range(2):
                    for buf in range(2):
                        out.append(f"v_add_u32 v{wr[2 * i + buf]}, {i * 128 * S_K * 4 + buf * self.stage_bytes}, {name}")
            else:
                out.append(f"v_mov_b32 v{wr[0]}, {name}")
        return out

    def text(self):
        body = ["s_nop 4"] + self.setup() + ["s_nop 1", "1:"] + self.step(0) + self.step(1)
        body += ["s_sub_u32 %[iters], %[iters], 1", "s_cmp_lg_u32 %[iters], 0", "s_cbranch_scc1 1b",
                 "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 7"]
        return body

    def name(self):
        return ("t" if self.ta else "n") + ("t" if self.tb else "n")


def emit(v):
    mi, ni = v.mi, v.ni
    lines = v.text()
    asm = "\n".join(f'        "{ln}\\n"' for piece in lines for ln in piece.split("\n"))
    accs = ", ".join(f'"+{{v[{v.acc(i, j)}:{v.acc(i, j) + 15}]}}"(acc[{i}][{j}])' for i in range(mi) for j in range(ni))
    fa0, fb0 = v.fa(0, 0, 0), v.fb(0, 0, 0)
    clob = ", ".join(f'"v{r}"' for r in list(range(v.frag_base + v.set_regs, v.frag_base + 2 * v.set_regs)) + list(range(v.first_temp, v.end_temp)))
    fa_t = "f32x8" if mi == 4 else "f32x4"
    fb_t = "f32x8" if ni == 4 else "f32x4"
    return f"""
// {'A transposed' if v.ta else 'A as stored'}, {'B transposed' if v.tb else 'B as stored'}: wavefront tile {32 * mi} x {32 * ni} ({mi} x {ni} matrix tiles),
// LDS buffer {v.stage_bytes} bytes, {v.end_temp} vector registers named in the block.
template <>
struct KLoopF32<{str(v.ta).lower()}, {str(v.tb).lower()}> {{
  static constexpr int MI = {mi}, NI = {ni};
  using fa_t = {fa_t};
  using fb_t = {fb_t};
  static __device__ __forceinline__ void run(f32x16 (&acc)[MI][NI], f32x4 (&sta)[2], f32x4 (&stb)[2], fa_t& fa, fb_t& fb,
                                             uint32_t goa0, uint32_t goa1, uint32_t gob0, uint32_t gob1, uint32_t rda,
                                             uint32_t rdb, uint32_t wra, uint32_t wrb, uint64_t base_a, uint64_t base_b,
                                             uint64_t step_a, uint64_t step_b, int iters) {{
    asm volatile(
{asm}
        : {accs},
          "+{{v[{v.st_a}:{v.st_a + 3}]}}"(sta[0]), "+{{v[{v.st_a + 4}:{v.st_a + 7}]}}"(sta[1]), "+{{v[{v.st_b}:{v.st_b + 3}]}}"(stb[0]), "+{{v[{v.st_b + 4}:{v.st_b + 7}]}}"(stb[1]),
          "+{{v[{fa0}:{fa0 + 2 * mi - 1}]}}"(fa), "+{{v[{fb0}:{fb0 + 2 * ni - 1}]}}"(fb),
          "+{{s[44:45]}}"(base_a), "+{{s[46:47]}}"(base_b), [iters] "+s"(iters)
        : [goa0] "v"(goa0), [goa1] "v"(goa1), [gob0] "v"(gob0), [gob1] "v"(gob1), [rda] "v"(rda), [rdb] "v"(rdb), [wra] "v"(wra),
          [wrb] "v"(wrb), "{{s[48:49]}}"(step_a), "{{s[50:51]}}"(step_b)
        : "memory", "scc", {clob});
  }}
}};
"""


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "sycl_blas_amd", "csrc", "gemm_f32_kloop.hpp")
    variants = [Variant(False, False, 4, 2), Variant(True, True, 4, 2), Variant(False, True, 2, 4), Variant(True, False, 2, 4)]
    with open(out, "w") as f:
        f.write("""// GENERATED by tools/gen_f32_kloop.py -- edit the generator, not this file.
//
// The steady-state K loop of the fp32 256 x 256 x 16 tile kernel (gemm_mfma_kernel<float, 256, 256, 16, ...>) as one
// block of gfx950 assembly per transpose combination: the role of compute_panel_gemm's main loop in the
// reference's tiled kernel (src/operations/blas3/gemm_local.hpp:451-479).  Every instruction of a K step has a
// fixed slot between the 64 matrix instructions of a wavefront; the matrix instructions run in the order of
// GemmBody::mma_group, so every C element keeps its k-ordered chain and the result keeps its bits.  One
// iteration of the block is two K steps (one out of each LDS buffer); the caller runs the first and the last
// steps of a tile, and everything at the edge of a matrix, through the C++ path.
//
// Vector registers are named in the block (the accumulators are v0..v127, two fragment sets, the staging
// registers of the tile in flight, the LDS bases); the operands of the asm statement are bound to exactly those
// registers.  Scalar registers s44..s51 carry the two global bases and their per-step increments.
#pragma once

#include <stdint.h>

namespace sbamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool TA, bool TB>
struct KLoopF32;
""")
        for v in variants:
            f.write(emit(v))
        f.write("\n}  // namespace sbamd\n")
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
