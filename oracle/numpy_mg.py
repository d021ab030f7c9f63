"""numpy_mg.py — INDEPENDENT numpy restatement of the 2D real Dirichlet path. TEST INFRASTRUCTURE ONLY
(like everything under oracle/): never imported by the product (evostencils_amd/).

Purpose ("pin the oracle"): a second, separately written restatement of what the
ExaStencils-generated executable computes for the statement list of
evostencils/code_generation/exastencils.py:684-925 on example_problems/Poisson/2D_FD_Poisson_fromL2.*,
with every [EXTERNAL] convention of SURVEY.md App. B behind a switch, so that
  (a) the C oracle (oracle/evomg_oracle.c) can be cross-checked against code that shares nothing with it
      (whole-array numpy expressions instead of node loops, dense batched solves instead of eliminations,
      plain np.sum reductions), and
  (b) the one numeric the reference records — notebooks/tutorial.ipynb:3373, rho = 0.9212764170210773 after
      100 iterations of the cycle at tutorial.ipynb:3421-3470 — can be searched for over the switch matrix
      (oracle/appb_matrix.py writes the table).

It interprets the same op program (include/evomg.h: evomg_op_t) as the HIP backend and the C oracle.
Arrays are indexed [i1][i0] (x fastest) with (2^L+1)^2 nodes including the boundary.

Switches (defaults = the conventions the C oracle and the HIP kernels implement):
  jacobi_update   "advance"       `with jacobi` writes the next slot, ONE advance makes it current (App. B.5)
                  "lost"          the update never becomes current (no advance, or two advances on a two-slot
                                  field: what exastencils.py:338-359 would produce if the debug layer 4 held two
                                  `[next]` lines per sweep)
  colour_order    (0, 1)          colour 0 = even i0+i1 counted from the boundary node first (App. B.7); or (1, 0)
  block_mode      "tiles"         non-overlapping tiles anchored at the first inner node (App. B.6, IR meaning)
                  "overlap_last"  a block at EVERY inner node, the last writer in loop order wins
                  "overlap_first" a block at every inner node, the first writer wins
  norm_nodes      "inner" | "all" residual norm over inner nodes or over all nodes (App. B.8)
  rhs             "analytic"      RHS_u as declared in the .exa2 (InitFields semantics kept)
                  "zero"          RHS_u left at zero (if stripping `Function InitFields`, exastencils.py:944,
                                  also lost the right-hand side initialisation)
  restriction     "fw"            [1/4 1/2 1/4]^2 (App. B.3); "fw4" = the unscaled transpose of the prolongation
  bc_after_write  True            Dirichlet values re-imposed after every field write (App. B.2); False: never
                                  (identical as long as no loop writes a boundary node — here for completeness)
  coarse          "cg"            CG, 1e-12 / 1000 iterations (App. B.9); "exact" = dense direct solve
"""
import numpy as np

from evostencils_amd import abi

DEFAULTS = dict(jacobi_update="advance", colour_order=(0, 1), block_mode="tiles", norm_nodes="inner",
                rhs="analytic", restriction="fw", bc_after_write=True, coarse="cg")


class NumpyMG:
    def __init__(self, problem, **switches):
        if problem.dimension != 2 or problem.scalar_kind != abi.REAL_F64 or problem.bc_kind != abi.BC_DIRICHLET:
            raise ValueError("numpy restatement covers the 2D real Dirichlet path")
        unknown = set(switches) - set(DEFAULTS)
        if unknown:
            raise ValueError(f"unknown switches {sorted(unknown)}")
        self.sw = dict(DEFAULTS, **switches)
        self.p = problem
        self.lo, self.hi = problem.min_level, problem.max_level
        self.st = {L: [float(np.real(c)) for c in problem.mg_stencil(L, problem.parameters)[:5]]
                   for L in range(self.lo, self.hi + 1)}
        self.reset()

    # ---- storage --------------------------------------------------------------------------------
    def n(self, L):
        return (1 << L) + 1

    def reset(self):
        self.f = {}
        for L in range(self.lo, self.hi + 1):
            z = lambda: np.zeros((self.n(L), self.n(L)))
            self.f[L] = {abi.F_SOL: z(), abi.F_RHS: z(), abi.F_RES: z(), abi.F_CGS_RHS: z()}
            if L == self.hi:
                self.f[L][abi.F_CORR] = z()
        self.f[self.hi][abi.F_SOL] = np.array(self.p.initial_solution(self.hi), dtype=np.float64)
        if self.sw["rhs"] == "analytic":
            self.f[self.hi][abi.F_RHS] = np.array(self.p.right_hand_side(self.hi), dtype=np.float64)
        self.bc = {(self.hi, abi.F_SOL): self.f[self.hi][abi.F_SOL].copy()}

    def canon(self, L, fld):
        return abi.F_SOL if (fld == abi.F_CORR and L < self.hi) else fld

    def get(self, L, fld):
        return self.f[L][self.canon(L, fld)]

    def put(self, L, fld, arr):
        """A field write: inner nodes from `arr`, boundary nodes keep / are reset to their Dirichlet values."""
        fld = self.canon(L, fld)
        old = self.f[L][fld]
        new = old.copy()
        new[1:-1, 1:-1] = arr[1:-1, 1:-1]
        if self.sw["bc_after_write"]:
            b = self.bc.get((L, fld))
            for sl in ((0, slice(None)), (-1, slice(None)), (slice(None), 0), (slice(None), -1)):
                new[sl] = b[sl] if b is not None else 0.0
        self.f[L][fld] = new

    # ---- operators ------------------------------------------------------------------------------
    def apply(self, L, x):
        c, w, e, s, nn = self.st[L]
        y = np.zeros_like(x)
        y[1:-1, 1:-1] = (c * x[1:-1, 1:-1] + w * x[1:-1, :-2] + e * x[1:-1, 2:] + s * x[:-2, 1:-1] + nn * x[2:, 1:-1])
        return y

    def residual(self, L, x, f):
        r = np.zeros_like(x)
        r[1:-1, 1:-1] = (f - self.apply(L, x))[1:-1, 1:-1]
        return r

    def off_diagonal(self, L, x):
        c, w, e, s, nn = self.st[L]
        y = np.zeros_like(x)
        y[1:-1, 1:-1] = w * x[1:-1, :-2] + e * x[1:-1, 2:] + s * x[:-2, 1:-1] + nn * x[2:, 1:-1]
        return y

    def point_star(self, L, x, f):
        """x* of the one-unknown local solve at every inner node."""
        xs = x.copy()
        xs[1:-1, 1:-1] = ((f - self.off_diagonal(L, x)) / self.st[L][0])[1:-1, 1:-1]
        return xs

    def restrict(self, fine):
        nc = (fine.shape[0] - 1) // 2 + 1
        c = np.zeros((nc, nc))
        w = {-1: 0.25, 0: 0.5, 1: 0.25}
        acc = np.zeros((nc - 2, nc - 2))
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                # coarse inner node (J, I) sits on fine node (2J, 2I)
                acc = acc + w[dy] * w[dx] * fine[2 + dy:2 * (nc - 2) + dy + 1:2, 2 + dx:2 * (nc - 2) + dx + 1:2]
        c[1:-1, 1:-1] = acc * (4.0 if self.sw["restriction"] == "fw4" else 1.0)
        return c

    def prolong(self, coarse):
        nf = 2 * (coarse.shape[0] - 1) + 1
        fz = np.zeros((nf, nf))
        fz[::2, ::2] = coarse
        fz[::2, 1::2] = 0.5 * (coarse[:, :-1] + coarse[:, 1:])
        fz[1::2, ::2] = 0.5 * (coarse[:-1, :] + coarse[1:, :])
        fz[1::2, 1::2] = 0.25 * (coarse[:-1, :-1] + coarse[:-1, 1:] + coarse[1:, :-1] + coarse[1:, 1:])
        return fz

    # ---- block local solves ---------------------------------------------------------------------------
    def _block_components(self, L, x, f, ax, ay, w, h):
        """Local solves of the w x h blocks anchored at the nodes (ay, ax) (index arrays, anchors such that
        the whole block is inner). Returns S[cy][cx] = array over anchors of the solution at cell (cx, cy)."""
        c0, cw, ce, cs, cn = self.st[L]
        offs = [(-1, 0, cw), (1, 0, ce), (0, -1, cs), (0, 1, cn)]
        cells = [(cx, cy) for cy in range(h) for cx in range(w)]
        A = np.zeros((len(cells), len(cells)))
        for a, (cx, cy) in enumerate(cells):
            A[a, a] = c0
            for dx, dy, v in offs:
                if 0 <= cx + dx < w and 0 <= cy + dy < h:
                    A[a, cells.index((cx + dx, cy + dy))] = v
        AY, AX = np.meshgrid(ay, ax, indexing="ij")
        rhs = np.zeros((len(cells),) + AY.shape)
        for a, (cx, cy) in enumerate(cells):
            g = f[AY + cy, AX + cx].copy()
            for dx, dy, v in offs:
                if not (0 <= cx + dx < w and 0 <= cy + dy < h):
                    g -= v * x[AY + cy + dy, AX + cx + dx]
            rhs[a] = g
        sol = np.linalg.solve(A, rhs.reshape(len(cells), -1)).reshape(rhs.shape)
        return {(cx, cy): sol[a] for a, (cx, cy) in enumerate(cells)}, AY, AX

    def block_star(self, L, x, f, bx, by):
        """x* of the block smoother under the selected block_mode (written into a copy of x)."""
        n = self.n(L)
        m = n - 2
        xs = x.copy()
        mode = self.sw["block_mode"]
        if mode == "tiles":
            # up to four rectangular families of tiles: full, cut in x, cut in y, cut in both
            fx, rx = divmod(m, bx)
            fy, ry = divmod(m, by)
            fam_x = [(1 + bx * np.arange(fx), bx)] + ([(np.array([1 + bx * fx]), rx)] if rx else [])
            fam_y = [(1 + by * np.arange(fy), by)] + ([(np.array([1 + by * fy]), ry)] if ry else [])
            for ax, w in fam_x:
                for ay, h in fam_y:
                    if len(ax) == 0 or len(ay) == 0:
                        continue
                    S, AY, AX = self._block_components(L, x, f, ax, ay, w, h)
                    for (cx, cy), v in S.items():
                        xs[AY + cy, AX + cx] = v
            return xs
        # overlapping: an anchor at every inner node; cells beyond the inner region drop out of the block
        order = []
        for w in range(1, min(bx, m) + 1):
            for h in range(1, min(by, m) + 1):
                # anchors whose available extent is exactly (w, h): all that fit the full block, or the one
                # column / row of anchors that sits w (h) nodes from the far boundary
                ax = np.arange(1, m - bx + 2) if w == bx else np.array([m - w + 1])
                ay = np.arange(1, m - by + 2) if h == by else np.array([m - h + 1])
                order.append(self._block_components(L, x, f, ax, ay, w, h))
        if mode == "overlap_last":
            # loop order i1 outer, i0 inner: the last anchor that covers a cell is the cell itself (offset 0, 0)
            for S, AY, AX in order:
                xs[AY, AX] = S[(0, 0)]
            return xs
        if mode == "overlap_first":
            # the first anchor that covers cell (i, j) is (max(1, i-bx+1), max(1, j-by+1))
            written = np.zeros((n, n), dtype=bool)
            # visit anchors in reverse so that earlier anchors overwrite later ones
            best = np.full((n, n), np.iinfo(np.int64).max, dtype=np.int64)
            for S, AY, AX in order:
                rank = AY.astype(np.int64) * n + AX
                for (cx, cy), v in S.items():
                    tgt = (AY + cy, AX + cx)
                    take = rank < best[tgt]
                    yy, xx = tgt[0][take], tgt[1][take]
                    xs[yy, xx] = v[take]
                    best[yy, xx] = rank[take]
                    written[yy, xx] = True
            return xs
        raise ValueError(mode)

    # ---- ops ------------------------------------------------------------------------------------
    def smooth(self, op):
        L, om = op.level, op.omega
        x, f = self.get(L, op.dst), self.get(L, op.src1)
        bx, by = op.block[0], op.block[1]
        if op.partitioning == abi.PART_RED_BLACK:
            if bx * by != 1:
                raise ValueError("red-black block smoothers are not part of the path")
            n = self.n(L)
            I1, I0 = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
            for colour in self.sw["colour_order"]:
                xs = self.point_star(L, x, f)
                mask = ((I0 + I1) % 2 == colour)
                new = np.where(mask, (1.0 - om) * x + om * xs, x)
                self.put(L, op.dst, new)
                x = self.get(L, op.dst)
            return
        if self.sw["jacobi_update"] == "lost":
            return
        xs = self.point_star(L, x, f) if bx * by == 1 else self.block_star(L, x, f, bx, by)
        self.put(L, op.dst, (1.0 - om) * x + om * xs)

    def coarse_solve(self):
        L = self.lo
        b = self.get(L, abi.F_CGS_RHS)
        x = self.get(L, abi.F_CORR).copy()
        if self.sw["coarse"] == "exact":
            m = self.n(L) - 2
            c0, cw, ce, cs, cn = self.st[L]
            idx = np.arange(m * m).reshape(m, m)
            A = np.zeros((m * m, m * m))
            A[idx, idx] = c0
            A[idx[:, 1:], idx[:, :-1]] = cw
            A[idx[:, :-1], idx[:, 1:]] = ce
            A[idx[1:, :], idx[:-1, :]] = cs
            A[idx[:-1, :], idx[1:, :]] = cn
            x[1:-1, 1:-1] = np.linalg.solve(A, b[1:-1, 1:-1].reshape(-1)).reshape(m, m)
            self.put(L, abi.F_CORR, x)
            return
        dot = lambda a, c: float(np.sum(a[1:-1, 1:-1] * c[1:-1, 1:-1]))
        r = self.residual(L, x, b)
        res0 = np.sqrt(dot(r, r))
        res = res0
        if res0 == 0.0:
            return
        pvec = r.copy()
        for _ in range(self.p.cgs_max_iterations):
            ap = self.apply(L, pvec)
            alpha = res * res / dot(pvec, ap)
            x = x + alpha * pvec
            r = r - alpha * ap
            nxt = np.sqrt(dot(r, r))
            if nxt <= self.p.cgs_target_res_reduction * res0:
                break
            beta = (nxt * nxt) / (res * res)
            pvec = r + beta * pvec
            res = nxt
        self.put(L, abi.F_CORR, x)

    def exec(self, op):
        L = op.level
        oc = op.opcode
        if oc == abi.OP_SMOOTH:
            self.smooth(op)
        elif oc == abi.OP_RESIDUAL:
            self.put(L, op.dst, self.residual(L, self.get(L, op.src0), self.get(L, op.src1)))
        elif oc == abi.OP_RICHARDSON:
            x = self.get(L, op.dst)
            self.put(L, op.dst, x + op.omega * self.residual(L, x, self.get(L, op.src1)))
        elif oc == abi.OP_RESTRICT:
            self.put(L, op.dst, self.restrict(self.get(L + 1, op.src0)))
        elif oc == abi.OP_PROLONG_ADD:
            self.put(L, op.dst, self.get(L, op.dst) + op.omega * self.prolong(self.get(L - 1, op.src0)))
        elif oc == abi.OP_PROLONG_SET:
            self.put(L, op.dst, self.prolong(self.get(L - 1, op.src0)))
        elif oc == abi.OP_SET_ZERO:
            self.put(L, op.dst, np.zeros((self.n(L), self.n(L))))
        elif oc == abi.OP_COPY:
            self.put(L, op.dst, self.get(L, op.src0))
        elif oc == abi.OP_COARSE_SOLVE:
            self.coarse_solve()
        else:
            raise ValueError(f"opcode {oc}")

    def run_ops(self, ops):
        for op in ops:
            self.exec(op)

    def norm(self, r):
        if self.sw["norm_nodes"] == "inner":
            r = r[1:-1, 1:-1]
        return float(np.sqrt(np.sum(r * r)))

    def solve(self, ops, max_iterations=None):
        """The generated solver loop (App. B.8). Returns the residual history [res0, res1, ...]."""
        self.reset()
        L = self.hi
        its = self.p.max_iterations if max_iterations is None else max_iterations
        r = self.residual(L, self.get(L, abi.F_SOL), self.get(L, abi.F_RHS))
        self.put(L, abi.F_RES, r)
        hist = [self.norm(self.get(L, abi.F_RES))]
        if hist[0] == 0.0:
            return np.array(hist)
        for _ in range(its):
            self.run_ops(ops)
            self.put(L, abi.F_RES, self.residual(L, self.get(L, abi.F_SOL), self.get(L, abi.F_RHS)))
            hist.append(self.norm(self.get(L, abi.F_RES)))
            if not np.isfinite(hist[-1]) or hist[-1] <= self.p.target_res_reduction * hist[0]:
                break
        return np.array(hist)


def convergence_factor(hist):
    """Geometric mean of the per-iteration ratios = what parse_output derives (exastencils.py:539-577)."""
    n = len(hist) - 1
    return float((hist[-1] / hist[0]) ** (1.0 / n)) if n > 0 and hist[0] > 0 else float("nan")
