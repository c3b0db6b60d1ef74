#!/usr/bin/env python3
"""Generate rl_brain_trainer_amd/csrc/kp1_fk_generated.inc: the fp32 handle's forward-kinematics chain as straight-line fp64 code specialised to
the robot constants of csrc/kp1_env.hip (ORIGIN_XYZ / ORIGIN_RPY / AXES_LOCAL, the reference's V51/ee_fk.py:14-61).

The generic chain (fk_chain in kp1_device.hpp) reads 3 + 3 + 6 x 27 + 15 folded constants through scalar loads and spends 18 + 27 + 9 fused
multiply-adds per joint.  Of the 162 entries of the folded K1 / Kc / Ks matrices only ~64 are larger than 1e-11 (the joint axes and origin
rotations are axis-aligned up to a 0.0101 rad tilt): with the constants as literals, the vanishing terms dropped and zeros propagated through the
products the chain is about half the arithmetic and loads nothing.  Dropping a term below 1e-11 moves a rotation entry by < 1e-11 and a position by
< 1e-11 m -- six orders below the fp32 handle's 1e-5 bar; the fp64 (strict) handle keeps the generic chain.

The folded constants are computed here with the arithmetic of fold_fk (same operation order, libm cos / sin) and written, unpruned, as FKG_CHECK:
kp1_create compares them bit for bit with what fold_fk computes at run time and refuses to start on a mismatch (regenerate with this script).

    python3 tools/gen_fk_chain.py            # rewrites the .inc
    python3 tools/gen_fk_chain.py --check    # exit 1 if the committed .inc is stale (tests/test_host_logic.py)
"""
import math
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "rl_brain_trainer_amd" / "csrc" / "kp1_env.hip"
OUT = ROOT / "rl_brain_trainer_amd" / "csrc" / "kp1_fk_generated.inc"
PRUNE = 1e-11


def table(text, name):
    m = re.search(r"const double " + name + r"\[7\]\[3\] = \{(.*?)\};", text, re.S)
    v = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(1))]
    assert len(v) == 21, name
    return [v[3 * i:3 * i + 3] for i in range(7)]


def mat3_mul(a, b):
    out = []
    for i in range(3):
        for j in range(3):
            s = 0.0
            for k in range(3):
                s += a[3 * i + k] * b[3 * k + j]
            out.append(s)
    return out


def rpy_to_rot(rpy):
    cr, sr, cp, sp, cy, sy = math.cos(rpy[0]), math.sin(rpy[0]), math.cos(rpy[1]), math.sin(rpy[1]), math.cos(rpy[2]), math.sin(rpy[2])
    rx = [1, 0, 0, 0, cr, -sr, 0, sr, cr]
    ry = [cp, 0, sp, 0, 1, 0, -sp, 0, cp]
    rz = [cy, -sy, 0, sy, cy, 0, 0, 0, 1]
    return mat3_mul(mat3_mul(rz, ry), rx)


def fold(text):
    """fold_fk<double> of kp1_env.hip, operation for operation"""
    xyz, rpy, axes = table(text, "ORIGIN_XYZ"), table(text, "ORIGIN_RPY"), table(text, "AXES_LOCAL")
    RA = [rpy_to_rot(r) for r in rpy]
    v0, p01 = [], []
    for r in range(3):
        v, pp = 0.0, xyz[0][r]
        for k in range(3):
            v += RA[0][3 * r + k] * axes[0][k]
            pp += RA[0][3 * r + k] * xyz[1][k]
        v0.append(v)
        p01.append(pp)
    k1, kc, ks, p = [], [], [], []
    for j in range(1, 7):
        ax = axes[j]
        nrm = math.sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-12
        x, y, z = ax[0] / nrm, ax[1] / nrm, ax[2] / nrm
        aat = [x * x, x * y, x * z, y * x, y * y, y * z, z * x, z * y, z * z]
        skew = [0, -z, y, z, 0, -x, -y, x, 0]
        eye_m = [(1.0 if e % 4 == 0 else 0.0) - aat[e] for e in range(9)]
        base = mat3_mul(RA[0], RA[1]) if j == 1 else list(RA[j])
        k1.append(mat3_mul(base, aat))
        kc.append(mat3_mul(base, eye_m))
        ks.append(mat3_mul(base, skew))
        if j >= 2:
            p.append(list(xyz[j]))
    return {"p01": p01, "v0": v0, "k1": k1, "kc": kc, "ks": ks, "p": p}


class Emit:
    """scalars are None (exact zero), a float (literal) or a str (variable name); fma / mul drop vanishing terms"""

    def __init__(self):
        self.lines, self.n = [], 0

    def lit(self, v):
        return None if abs(v) < PRUNE else float(v)

    @staticmethod
    def txt(v):
        return v if isinstance(v, str) else float(v).hex()

    def new(self, expr):
        self.n += 1
        name = f"t{self.n}"
        self.lines.append(f"  const double {name} = {expr};")
        return name

    def mul(self, a, b):
        if a is None or b is None:
            return None
        if isinstance(a, float) and isinstance(b, float):
            return a * b
        for x, y in ((a, b), (b, a)):
            if isinstance(x, float) and x == 1.0:
                return y                                   # 1.0 * y is y exactly
        return self.new(f"{self.txt(a)} * {self.txt(b)}")

    def fma(self, a, b, c):
        if a is None or b is None:
            return c
        if c is None:
            return self.mul(a, b)
        if isinstance(a, float) and isinstance(b, float) and isinstance(c, float):
            return a * b + c
        return self.new(f"__builtin_fma({self.txt(a)}, {self.txt(b)}, {self.txt(c)})")


def generate(text):
    K = fold(text)
    e = Emit()
    L = e.lit
    e.lines.append("  double s, c;")
    p = [e.fma(L(K["v0"][i]), "q[0]", L(K["p01"][i])) for i in range(3)]
    Rm = None
    for j in range(1, 7):
        m = j - 1
        e.lines.append(f"  kp_sincos_kin(q[{j}], &s, &c);")
        sj, cj = e.new("s"), e.new("c")
        D = [e.fma(sj, L(K["ks"][m][k]), e.fma(cj, L(K["kc"][m][k]), L(K["k1"][m][k]))) for k in range(9)]
        if j == 1:
            Rm = D
            continue
        P = [L(v) for v in K["p"][j - 2]]
        p = [e.fma(Rm[3 * r + 2], P[2], e.fma(Rm[3 * r + 1], P[1], e.fma(Rm[3 * r + 0], P[0], p[r]))) for r in range(3)]
        Rm = [e.fma(Rm[3 * r + 2], D[6 + cc], e.fma(Rm[3 * r + 1], D[3 + cc], e.mul(Rm[3 * r + 0], D[cc]))) for r in range(3) for cc in range(3)]
    for i in range(3):
        e.lines.append(f"  p[{i}] = {e.txt(p[i]) if p[i] is not None else '0.0'};")
    for k in range(9):
        e.lines.append(f"  Rm[{k}] = {e.txt(Rm[k]) if Rm[k] is not None else '0.0'};")
    flat = K["p01"] + K["v0"] + [v for mm in K["k1"] for v in mm] + [v for mm in K["kc"] for v in mm] + [v for mm in K["ks"] for v in mm] + [v for pp in K["p"] for v in pp]
    kept = sum(1 for name in ("k1", "kc", "ks") for mm in K[name] for v in mm if abs(v) >= PRUNE)
    n_fma = sum(1 for ln in e.lines if "__builtin_fma" in ln or " * " in ln)
    head = f"""// kp1_fk_generated.inc -- GENERATED by tools/gen_fk_chain.py from the robot constants of kp1_env.hip; do not edit.
// The fp32 handle's FK chain (fk_pose6_kin<float>) as straight-line fp64 code: {kept} of the 162 folded K1 / Kc / Ks entries are >= {PRUNE:g} and kept as
// literals, the rest dropped with their products ({n_fma} multiply-adds instead of 18 + 5 x 54 + 3).  FKG_CHECK = the unpruned folded constants in
// DevFk<double> member order (p01, v0, k1, kc, ks, p): kp1_create compares them bit for bit with fold_fk's and fails on a mismatch.
"""
    check = "constexpr double FKG_CHECK[%d] = {\n" % len(flat) + "".join(f"    {float(v).hex()},\n" for v in flat) + "};\n"
    body = ("__device__ __forceinline__ void fk_chain_generated(const double* __restrict__ q, double* __restrict__ p, double* __restrict__ Rm) {\n"
            "#pragma clang fp contract(off)\n" + "\n".join(e.lines) + "\n}\n")
    return head + check + body


def main():
    text = SRC.read_text()
    new = generate(text)
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != new:
            print("kp1_fk_generated.inc is stale: run python3 tools/gen_fk_chain.py", file=sys.stderr)
            sys.exit(1)
        return
    OUT.write_text(new)
    print("wrote", OUT, len(new.splitlines()), "lines")


if __name__ == "__main__":
    main()
