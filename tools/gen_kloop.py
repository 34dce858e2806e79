#!/usr/bin/env python3
"""Writes sycl_blas_amd/csrc/gemm_kloop_asm.hpp: the steady-state K loop of the fp32 / fp64 tile kernels
(gemm_mfma_kernel in gemm_mfma.hpp) as one block of gfx950 assembly per tile shape and transpose combination.

Why assembly: in the compiled loop hipcc decides where the LDS reads, LDS writes, global loads and their waits
sit between the matrix instructions (a scheduling pattern only orders what it claims), and it spends vector
instructions on LDS / global addresses every step.  Here every instruction of a K step has a fixed slot:

  * the matrix instructions of a wavefront in the order of GemmBody::mma_group -- for every C element k
    ascends, so the result keeps the bits of the compiled loop (the tests compare whole buffers);
  * the fragments of group g + GA are requested behind the first matrix instructions of group g (GA = 1 with
    two register sets for wavefront tiles of several matrix tiles, GA = 2 with four sets for a single one) and
    waited for with a COUNTED s_waitcnt right in front of the group that needs them;
  * the tile that arrived in registers during the previous step is written to the other LDS buffer and the next
    tile requested into the registers just freed, vector by vector, spread over the groups in front of the
    barrier; the wait for each vector is counted too (the loads of the previous step return in order);
  * all LDS addresses are a per-lane base register plus an immediate (the two buffers are the two halves of an
    unrolled pair of steps), all global addresses a scalar base plus a per-lane 32-bit offset: no address
    arithmetic on the vector unit inside the loop;
  * one s_barrier per step, in front of the last GA groups, as in the compiled loop.

The LDS images are exactly those of TileLoader / GemmTile in gemm_mfma.hpp, so the prologue, the last steps, the
K tail and the epilogue stay in C++ around the block.
Run from the repository root:  python tools/gen_kloop.py
"""
import os


class Operand:
    def __init__(self, cfg, mn, kmajor):
        self.mn, self.kmajor = mn, kmajor
        self.S = cfg.bk + 2 if kmajor else mn + cfg.pad_mn
        elems = mn * self.S if kmajor else cfg.bk * self.S
        self.elems = (elems + 3) & ~3
        self.nv = mn * cfg.bk // cfg.V // cfg.NT
        assert self.nv in (1, 2, 3, 4) and mn * cfg.bk % (cfg.V * cfg.NT) == 0
        self.per_row = cfg.bk // cfg.V if kmajor else mn // cfg.V
        # rows of the image between a thread's vectors i and i + 1 -- unless a row's vector count does not divide
        # the workgroup (96 and 192 wide M/N-contiguous tiles): then every vector has a per-lane LDS address of its own
        self.per_lane_wr = cfg.NT % self.per_row != 0
        self.rows_per_vec = cfg.NT // self.per_row
        self.nw = self.nv if self.per_lane_wr else 1  # write addresses the caller passes


class Cfg:
    def __init__(self, dtype, bm, bn, bk, wm, wn, ta, tb, ga, small=False):
        """small: fp32 on v_mfma_f32_16x16x4_f32 (MfmaF32Small in gemm_mfma.hpp: the 96 x 96 and 192 x 192 tiles)"""
        self.dtype, self.bm, self.bn, self.bk, self.wm, self.wn, self.ta, self.tb, self.ga = dtype, bm, bn, bk, wm, wn, ta, tb, ga
        self.small = small
        f32 = dtype == "float"
        assert f32 or not small
        self.esz = 4 if f32 else 8
        self.V = 16 // self.esz
        self.T = 32 if f32 and not small else 16              # edge of a matrix tile
        self.accr = 4 if small else (16 if f32 else 8)        # registers of one accumulator tile
        self.gs = 2 if f32 and not small else 1               # matrix instructions per tile and fragment group
        self.fregs = 1 if small else 2                        # registers of one tile's fragment of a group
        self.pad_mn = 16 if small or not f32 else 0
        self.NT = wm * wn * 64
        self.mi, self.ni = bm // wm // self.T, bn // wn // self.T
        self.ngrp = bk // 4
        self.nsets = 2 if ga == 1 else 4
        assert self.ngrp % self.nsets == 0 and ga < self.nsets
        self.A = Operand(self, bm, ta)
        self.B = Operand(self, bn, not tb)
        self.a_bytes = self.A.elems * self.esz
        self.stage = (self.A.elems + self.B.elems) * self.esz
        # ---- register map
        self.nacc = self.mi * self.ni * self.accr
        # accumulator registers per asm operand: the largest register-tuple size that divides them
        # (power-of-two tuples only: hipcc keeps a 12-element vector in scratch memory)
        self.chunk = next(c for c in (32, 16, 8, 4) if self.nacc % c == 0)
        # the fragments of A and of B of one set are asm operands too: power-of-two tuples, which also keeps every
        # tuple at an even register as this part requires
        pow2 = lambda n: 1 << (n - 1).bit_length()  # noqa: E731
        self.fa_w, self.fb_w = pow2(self.fregs * self.mi), pow2(self.fregs * self.ni)
        self.fb_off = self.fa_w
        self.set_regs = max(2, self.fa_w + self.fb_w)
        self.sta_w, self.stb_w = pow2(4 * self.A.nv), pow2(4 * self.B.nv)
        self.frag0 = self.nacc
        self.st_a = self.frag0 + self.nsets * self.set_regs
        self.st_b = self.st_a + self.sta_w
        # the per-lane inputs, one register tuple: global offsets of A's and B's vectors, read bases, write bases
        self.lane0 = self.st_b + self.stb_w
        names = ([f"goa{i}" for i in range(self.A.nv)] + [f"gob{i}" for i in range(self.B.nv)] + ["rda", "rdb"] +
                 [f"wra{i}" for i in range(self.A.nw)] + [f"wrb{i}" for i in range(self.B.nw)])
        self.lane_n = next(c for c in (4, 8, 16) if c >= len(names))
        self.lane = {nm: self.lane0 + q for q, nm in enumerate(names)}
        self.lane_names = names
        self.temp0 = self.lane0 + self.lane_n
        self.temps = []  # (register, setup instruction)
        nxt = self.temp0
        self.rd = {}
        self.wr = {}
        for name, op, ntiles in (("a", self.A, self.mi), ("b", self.B, self.ni)):
            nxt = self.alloc_bases(name, op, ntiles, nxt)
        self.end = nxt
        assert self.end <= 250, self.end

    def alloc_bases(self, name, op, ntiles, nxt):
        """Per-lane LDS base registers derived from the bases the caller passes: reads rd[name][(tile, buf)],
        writes wr[name][(vector, buf)].  A register is shared wherever an immediate can carry the difference."""
        f32 = self.dtype == "float"
        rd, wr = {}, {}
        rd_in, wr_in = f"v{self.lane['rd' + name]}", f"v{self.lane['wr' + name + '0']}"
        if self.small:
            # ds_read_b32 with a 16-bit byte offset
            tile_step = 16 * op.S * 4 if op.kmajor else 16 * 4
            grp_step = 16 if op.kmajor else 4 * op.S * 4
            top = (ntiles - 1) * tile_step + (self.ngrp - 1) * grp_step + self.stage
            per_buf = top > 65535
            for buf in range(2):
                if buf == 0:
                    r = self.lane["rd" + name]
                elif not per_buf:
                    r = rd[(0, 0)]
                else:
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {self.stage}, {rd_in}"))
                    r = nxt
                    nxt += 1
                for t in range(ntiles):
                    rd[(t, buf)] = r
            self.rd_buf_imm = getattr(self, "rd_buf_imm", {})
            self.rd_buf_imm[name] = not per_buf
        elif f32 and not op.kmajor:
            # ds_read2st64_b32: offsets in units of 256 B, 8 bits -> tiles (128 B apart) need their own base
            assert op.S * 4 % 256 == 0 and self.stage % 256 == 0
            top = (4 * (self.ngrp - 1) + 2) * (op.S * 4 // 256) + self.stage // 256
            per_buf = top > 255
            for t in range(ntiles):
                for buf in range(2):
                    if buf == 1 and not per_buf:
                        rd[(t, 1)] = rd[(t, 0)]
                        continue
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {128 * t + buf * self.stage}, {rd_in}"))
                    rd[(t, buf)] = nxt
                    nxt += 1
            self.rd_buf_imm = getattr(self, "rd_buf_imm", {})
            self.rd_buf_imm[name] = not per_buf
        else:
            # ds_read_b64 with a 16-bit byte offset
            tile_step = (32 if f32 else 16) * op.S * self.esz if op.kmajor else 16 * 8
            grp_step = 16 if (f32 and op.kmajor) else (32 if op.kmajor else 4 * op.S * 8)
            top = (ntiles - 1) * tile_step + (self.ngrp - 1) * grp_step + self.stage
            per_buf = top > 65535
            for buf in range(2):
                if buf == 1 and not per_buf:
                    r = rd[(0, 0)]
                else:
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {buf * self.stage}, {rd_in}"))
                    r = nxt
                    nxt += 1
                for t in range(ntiles):
                    rd[(t, buf)] = r
            self.rd_buf_imm = getattr(self, "rd_buf_imm", {})
            self.rd_buf_imm[name] = not per_buf
        self.rd[name] = rd
        vec_step = op.rows_per_vec * op.S * self.esz
        if op.per_lane_wr:
            # a per-lane address for every vector (passed in); the other buffer by immediate where it fits
            per_buf = self.stage + 16 > 65535
            for i in range(op.nv):
                wr[(i, 0)] = self.lane[f"wr{name}{i}"]
                if per_buf:
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {self.stage}, v{wr[(i, 0)]}"))
                    wr[(i, 1)] = nxt
                    nxt += 1
                else:
                    wr[(i, 1)] = wr[(i, 0)]
            self.wr_imm = getattr(self, "wr_imm", {})
            self.wr_imm[name] = not per_buf
        elif f32 and op.kmajor and not self.small:
            # ds_write2_b32: 8-bit offsets only -> one base per (vector, buffer)
            for i in range(op.nv):
                for buf in range(2):
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {i * vec_step + buf * self.stage}, {wr_in}"))
                    wr[(i, buf)] = nxt
                    nxt += 1
            self.wr_imm = getattr(self, "wr_imm", {})
            self.wr_imm[name] = None
        else:
            top = (op.nv - 1) * vec_step + self.stage
            per_buf = top > 65535
            for buf in range(2):
                if buf == 1 and not per_buf:
                    r = wr[(0, 0)]
                else:
                    self.temps.append((nxt, f"v_add_u32 v{nxt}, {buf * self.stage}, {wr_in}"))
                    r = nxt
                    nxt += 1
                for i in range(op.nv):
                    wr[(i, buf)] = r
            self.wr_imm = getattr(self, "wr_imm", {})
            self.wr_imm[name] = not per_buf
        self.wr[name] = wr
        return nxt

    # ---- registers
    def acc(self, i, j):
        return (i * self.ni + j) * self.accr

    def frag(self, s, name, t):
        return self.frag0 + s * self.set_regs + (self.fregs * t if name == "a" else self.fb_off + self.fregs * t)

    # ---- instruction text
    def reads(self, g, buf):
        """fragment reads of group g (of the tile in LDS buffer buf) into set g % nsets"""
        s = g % self.nsets
        f32 = self.dtype == "float"
        out = []
        for name, op, ntiles in (("a", self.A, self.mi), ("b", self.B, self.ni)):
            for t in range(ntiles):
                d = self.frag(s, name, t)
                base = self.rd[name][(t, buf)]
                bufoff = buf * self.stage if self.rd_buf_imm[name] else 0
                if self.small:
                    off = (t * 16 * op.S + 4 * g) * 4 if op.kmajor else (4 * g * op.S + t * 16) * 4
                    out.append(("lds", f"ds_read_b32 v{d}, v{base} offset:{off + bufoff}"))
                elif f32 and not op.kmajor:
                    unit = op.S * 4 // 256
                    o0 = 4 * g * unit + bufoff // 256
                    out.append(("lds", f"ds_read2st64_b32 v[{d}:{d + 1}], v{base} offset0:{o0} offset1:{o0 + 2 * unit}"))
                elif f32:
                    out.append(("lds", f"ds_read_b64 v[{d}:{d + 1}], v{base} offset:{t * 32 * op.S * 4 + g * 16 + bufoff}"))
                elif op.kmajor:
                    out.append(("lds", f"ds_read_b64 v[{d}:{d + 1}], v{base} offset:{t * 16 * op.S * 8 + g * 32 + bufoff}"))
                else:
                    out.append(("lds", f"ds_read_b64 v[{d}:{d + 1}], v{base} offset:{(4 * g * op.S + t * 16) * 8 + bufoff}"))
        return out

    def moves(self, buf):
        """per staged vector: [write it to LDS buffer buf, request its successor]; the wait in front of each
        write is filled in by the counting pass"""
        out = []
        f32 = self.dtype == "float"
        for name, op, st, goff, sbase in (("a", self.A, self.st_a, "goa", "s[44:45]"), ("b", self.B, self.st_b, "gob", "s[46:47]")):
            for i in range(op.nv):
                r = st + 4 * i
                base = self.wr[name][(i, buf)]
                piece = [("vmwait", (name, i))]
                if op.per_lane_wr:
                    piece.append(("lds", f"ds_write_b128 v{base}, v[{r}:{r + 3}] offset:{buf * self.stage if self.wr_imm[name] else 0}"))
                elif self.small and op.kmajor:
                    # rows of 34 floats: 8-byte aligned only
                    off = i * op.rows_per_vec * op.S * 4 + (buf * self.stage if self.wr_imm[name] else 0)
                    piece.append(("lds", f"ds_write_b64 v{base}, v[{r}:{r + 1}] offset:{off}"))
                    piece.append(("lds", f"ds_write_b64 v{base}, v[{r + 2}:{r + 3}] offset:{off + 8}"))
                elif f32 and op.kmajor:
                    piece.append(("lds", f"ds_write2_b32 v{base}, v{r}, v{r + 2} offset1:1"))
                    piece.append(("lds", f"ds_write2_b32 v{base}, v{r + 1}, v{r + 3} offset0:2 offset1:3"))
                else:
                    off = i * op.rows_per_vec * op.S * self.esz + (buf * self.stage if self.wr_imm[name] else 0)
                    piece.append(("lds", f"ds_write_b128 v{base}, v[{r}:{r + 3}] offset:{off}"))
                piece.append(("vmem", (name, i), f"global_load_dwordx4 v[{r}:{r + 3}], v{self.lane[goff + str(i)]}, {sbase}"))
                out.append(piece)
        return out

    def mfmas(self, g):
        s = g % self.nsets
        f32 = self.dtype == "float"
        out = []
        for e in range(self.gs):
            for i in range(self.mi):
                for j in range(self.ni):
                    a = self.acc(i, j)
                    if self.small:
                        out.append(f"v_mfma_f32_16x16x4_f32 v[{a}:{a + 3}], v{self.frag(s, 'b', j)}, v{self.frag(s, 'a', i)}, v[{a}:{a + 3}]")
                    elif f32:
                        out.append(f"v_mfma_f32_32x32x2_f32 v[{a}:{a + 15}], v{self.frag(s, 'b', j) + e}, v{self.frag(s, 'a', i) + e}, v[{a}:{a + 15}]")
                    else:
                        fb, fa = self.frag(s, "b", j), self.frag(s, "a", i)
                        out.append(f"v_mfma_f64_16x16x4_f64 v[{a}:{a + 7}], v[{fb}:{fb + 1}], v[{fa}:{fa + 1}], v[{a}:{a + 7}]")
        return out

    def step_items(self, cur):
        """One K step out of buffer cur as a list of items: ('mfma', text) / ('lds', text) / ('vmem', key, text) /
        ('vmwait', key) / ('need', group) / ('barrier',) / ('salu', text)."""
        nxt = cur ^ 1
        items = []
        moves = self.moves(nxt)
        pre_groups = self.ngrp - self.ga  # groups in front of the barrier: all the tile traffic goes there
        flat = [it for piece in moves for it in piece]
        flat.append(("salu", "s_add_u32 s44, s44, s48\ns_addc_u32 s45, s45, s49\ns_add_u32 s46, s46, s50\ns_addc_u32 s47, s47, s51"))
        nm = self.mi * self.ni * self.gs
        # spread the traffic evenly over the matrix instructions in front of the barrier (behind each group's reads)
        slots_total = pre_groups * nm
        traffic_at = {}
        for q, it in enumerate(flat):
            slot = min(slots_total - 1, q * slots_total // len(flat))
            traffic_at.setdefault(slot, []).append(it)
        for g in range(self.ngrp):
            items.append(("need", g))
            if g == pre_groups:
                items.append(("barrier",))
            tgt = g + self.ga
            reads = self.reads(tgt, cur) if tgt < self.ngrp else self.reads(tgt - self.ngrp, nxt)
            m = self.mfmas(g)
            per = -(-len(reads) // nm)
            for n, ins in enumerate(m):
                items.append(("mfma", ins))
                items += reads[n * per:(n + 1) * per]
                if g < pre_groups:
                    items += traffic_at.get(g * nm + n, [])
        return items

    def body(self):
        """Two steps (buffers 0 and 1) with every wait counted over the periodic instruction stream."""
        period = self.step_items(0) + self.step_items(1)
        stream = period * 3  # steady state: take the waits of the middle period
        lds_issue = []       # index in `stream` of every LDS instruction
        vm_issue = {}        # key -> stream index of its latest request
        vm_order = []
        read_of_group = {}   # (step parity, group) handled implicitly: reads carry their target through position
        # Which LDS reads serve which ('need', g)?  The reads of group g of the tile computed in step p are the
        # ones issued GA groups earlier; find them by walking backwards to the latest reads into set g % nsets.
        out_text = {}
        lds_count = 0
        vm_count = 0
        lds_pos = []   # for every stream item: number of LDS instructions issued before it
        vm_pos = []
        for it in stream:
            lds_pos.append(lds_count)
            vm_pos.append(vm_count)
            if it[0] == "lds":
                lds_count += 1
            elif it[0] == "vmem":
                vm_count += 1
        n = len(period)
        waits = {}
        for idx in range(n, 2 * n):
            it = stream[idx]
            if it[0] == "need":
                g = it[1]
                s = g % self.nsets
                regs = set()
                for name, ntiles in (("a", self.mi), ("b", self.ni)):
                    for t in range(ntiles):
                        d0 = self.frag(s, name, t)
                        regs.add(f"v{d0}" if self.fregs == 1 else f"v[{d0}:{d0 + 1}]")
                # latest LDS read into one of these registers before idx
                last = None
                for j in range(idx - 1, -1, -1):
                    jt = stream[j]
                    if jt[0] == "lds" and jt[1].startswith("ds_read") and jt[1].split()[1].rstrip(",") in regs:
                        last = j
                        break
                assert last is not None
                outstanding_after = lds_pos[idx] - (lds_pos[last] + 1)
                waits[idx - n] = f"s_waitcnt lgkmcnt({min(15, outstanding_after)})"
            elif it[0] == "vmwait":
                key = it[1]
                last = None
                for j in range(idx - 1, -1, -1):
                    jt = stream[j]
                    if jt[0] == "vmem" and jt[1] == key:
                        last = j
                        break
                assert last is not None
                outstanding_after = vm_pos[idx] - (vm_pos[last] + 1)
                waits[idx - n] = f"s_waitcnt vmcnt({min(63, outstanding_after)})"
        lines = []
        for k, it in enumerate(period):
            if it[0] in ("need", "vmwait"):
                lines.append(waits[k])
            elif it[0] == "barrier":
                lines += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
            elif it[0] == "vmem":
                lines.append(it[2])
            else:
                lines.append(it[1])
        # a wait directly followed by the barrier's full drain is redundant
        cleaned = []
        for k, ln in enumerate(lines):
            if ln.startswith("s_waitcnt lgkmcnt(") and k + 1 < len(lines) and lines[k + 1] == "s_waitcnt lgkmcnt(0)":
                continue
            cleaned.append(ln)
        return cleaned

    def text(self):
        setup = [ins for _, ins in self.temps]
        return (["s_nop 4"] + setup + ["s_nop 1", "1:"] + self.body() +
                ["s_sub_u32 %[iters], %[iters], 1", "s_cmp_lg_u32 %[iters], 0", "s_cbranch_scc1 1b",
                 "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 7"])

    def ctype(self, regs):
        n = regs * 4 // self.esz
        return f"{self.dtype} __attribute__((ext_vector_type({n})))" if n > 1 else self.dtype


def emit(c):
    lines = c.text()
    asm = "\n".join(f'        "{ln}\\n"' for piece in lines for ln in piece.split("\n"))
    nch = c.nacc // c.chunk
    ops = [f'"+{{v[{q * c.chunk}:{q * c.chunk + c.chunk - 1}]}}"(acc[{q}])' for q in range(nch)]
    ops.append(f'"+{{v[{c.st_a}:{c.st_a + c.sta_w - 1}]}}"(sta)')
    ops.append(f'"+{{v[{c.st_b}:{c.st_b + c.stb_w - 1}]}}"(stb)')
    for s in range(c.ga):
        fa, fb = c.frag(s, "a", 0), c.frag(s, "b", 0)
        ops.append(f'"+{{v[{fa}:{fa + c.fa_w - 1}]}}"(fa[{s}])')
        ops.append(f'"+{{v[{fb}:{fb + c.fb_w - 1}]}}"(fb[{s}])')
    ops += ['"+{s[44:45]}"(base_a)', '"+{s[46:47]}"(base_b)', '[iters] "+s"(iters)']
    ins = [f'"{{v[{c.lane0}:{c.lane0 + c.lane_n - 1}]}}"(lanes)', '"{s[48:49]}"(step_a)', '"{s[50:51]}"(step_b)']
    assert len(ops) + len(ins) <= 30, len(ops) + len(ins)
    clob_regs = list(range(c.frag0 + c.ga * c.set_regs, c.st_a)) + [r for r, _ in c.temps]
    clob = ", ".join(f'"v{r}"' for r in clob_regs)
    t = lambda b: "true" if b else "false"  # noqa: E731
    return f"""
// {c.dtype} {c.bm} x {c.bn} x {c.bk} on {c.wm} x {c.wn} wavefronts, {'A transposed' if c.ta else 'A as stored'}, {'B transposed' if c.tb else 'B as stored'}:
// wavefront tile of {c.mi} x {c.ni} matrix tiles, fragments {c.ga} group(s) ahead, LDS buffer {c.stage} bytes,
// vector registers v0..v{c.end - 1} named in the block.
template <>
struct KLoop<{c.dtype}, {c.bm}, {c.bn}, {c.bk}, {c.wm}, {c.wn}, {t(c.ta)}, {t(c.tb)}> {{
  static constexpr bool available = true, SMALL = {t(c.small)};  // SMALL: the 16x16x4 fp32 matrix instruction
  static constexpr int MI = {c.mi}, NI = {c.ni}, GA = {c.ga}, NVA = {c.A.nv}, NVB = {c.B.nv};
  // the per-lane inputs in one register tuple: global byte offsets of the lane's vectors of A and of B, LDS read
  // bases, LDS write addresses (one per operand, or one per vector where a row's vectors do not divide the workgroup)
  static constexpr int NWA = {c.A.nw}, NWB = {c.B.nw}, LANE_N = {c.lane_n};
  static constexpr int L_GOA = {c.lane['goa0'] - c.lane0}, L_GOB = {c.lane['gob0'] - c.lane0}, L_RDA = {c.lane['rda'] - c.lane0}, L_RDB = {c.lane['rdb'] - c.lane0}, L_WRA = {c.lane['wra0'] - c.lane0}, L_WRB = {c.lane['wrb0'] - c.lane0};
  typedef uint32_t __attribute__((ext_vector_type({c.lane_n}))) lane_t;
  static constexpr int ACC_CHUNKS = {nch}, CHUNK = {c.chunk * 4 // c.esz};  // accumulator elements per asm operand
  typedef {c.ctype(c.chunk)} acc_t;
  typedef {c.ctype(c.sta_w)} sta_t;
  typedef {c.ctype(c.stb_w)} stb_t;
  typedef {c.ctype(c.fa_w)} fa_t;
  typedef {c.ctype(c.fb_w)} fb_t;
  static __device__ __forceinline__ void run(acc_t (&acc)[ACC_CHUNKS], sta_t& sta, stb_t& stb, fa_t (&fa)[GA], fb_t (&fb)[GA],
                                             lane_t lanes, uint64_t base_a, uint64_t base_b, uint64_t step_a, uint64_t step_b,
                                             int iters) {{
    asm volatile(
{asm}
        : {", ".join(ops)}
        : {", ".join(ins)}
        : "memory", "scc", {clob});
  }}
}};
"""


CONFIGS = []
for ta, tb in ((False, False), (True, True)):
    CONFIGS.append(Cfg("float", 256, 256, 16, 2, 4, ta, tb, 1))
for ta, tb in ((False, True), (True, False)):
    CONFIGS.append(Cfg("float", 256, 256, 16, 4, 2, ta, tb, 1))
for ta in (False, True):
    for tb in (False, True):
        CONFIGS.append(Cfg("double", 256, 128, 16, 4, 2, ta, tb, 1))
        # the smaller fp32 tiles: 128 x 128 x 32 on eight wavefronts of 32 x 64; 64 x 64 on four wavefronts of one
        # matrix tile each (fragments two groups ahead), K steps of 32 and 64
        CONFIGS.append(Cfg("float", 128, 128, 32, 4, 2, ta, tb, 1))
        CONFIGS.append(Cfg("float", 64, 64, 32, 2, 2, ta, tb, 2))
        CONFIGS.append(Cfg("float", 64, 64, 64, 2, 2, ta, tb, 2))
        # the smaller fp64 tiles: 128 x 128 x 16 and 64 x 64 x 16 on four wavefronts (16 / 4 matrix tiles each)
        CONFIGS.append(Cfg("double", 128, 128, 16, 2, 2, ta, tb, 1))
        CONFIGS.append(Cfg("double", 64, 64, 16, 2, 2, ta, tb, 1))
        # the 96 x 96 and 192 x 192 fp32 tiles on the 16x16x4 matrix instruction: wavefront tiles of 3 x 3 and 6 x 3
        CONFIGS.append(Cfg("float", 96, 96, 32, 2, 2, ta, tb, 1, small=True))
        CONFIGS.append(Cfg("float", 192, 192, 32, 2, 4, ta, tb, 1, small=True))
        # the same two tiles in fp64 (K step 16)
        CONFIGS.append(Cfg("double", 96, 96, 16, 2, 2, ta, tb, 1))
        CONFIGS.append(Cfg("double", 192, 192, 16, 2, 4, ta, tb, 1))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "sycl_blas_amd", "csrc", "gemm_kloop_asm.hpp")
    with open(out, "w") as f:
        f.write("""// GENERATED by tools/gen_kloop.py -- edit the generator, not this file.
//
// The steady-state K loop of the fp32 / fp64 tile kernels (gemm_mfma_kernel) as one block of gfx950 assembly per
// tile shape and transpose combination: the role of compute_panel_gemm's main loop in the reference's tiled
// kernel (src/operations/blas3/gemm_local.hpp:451-479).  Every instruction of a K step has a fixed slot between
// the matrix instructions of a wavefront, every wait is counted; the matrix instructions run in the order of
// GemmBody::mma_group, so every C element keeps its k-ordered chain and the result keeps its bits.  One iteration
// of a block is two K steps (one out of each LDS buffer); the caller runs the first and the last steps of a tile,
// the K tail, and everything at the edge of a matrix, through the C++ path.
//
// Vector registers are named in a block (accumulators from v0, the fragment sets, the staging registers of the
// tile in flight, the LDS bases); the operands of the asm statement are bound to exactly those registers.
// Scalar registers s44..s51 carry the two global bases and their per-step increments.
#pragma once

#include <stdint.h>

namespace sbamd {

template <typename T, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB>
struct KLoop {
  static constexpr bool available = false, SMALL = false;
  static constexpr int MI = 0, NI = 0, GA = 0;
};
""")
        for c in CONFIGS:
            f.write(emit(c))
        f.write("\n}  // namespace sbamd\n")
    print("wrote", os.path.normpath(out), "with", len(CONFIGS), "blocks")


if __name__ == "__main__":
    main()
