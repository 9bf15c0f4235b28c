"""Static checks on the gfx950 code that SHIPS: the code objects inside a built .so (marbler_amd/librobogym_hip.so), not a
re-compile at other flags.  Used by tests/test_kernel_resources.py (CPU tier, ~10 s) and as a CLI:

    python tools/isa_scan.py [path/to/lib.so]        # every kernel: resources, DOT hazard, exec-prologue check

Three checks, each born from an incident of this repo:

* `dot_hazards`   -- gfx940 / gfx950 do not interlock a VALU access to the destination of a DOT instruction within three wait
                     states; the inline-asm `v_dot2_f32_f16` of csrc/device_common.h is invisible to the compiler's hazard
                     recogniser (round 3: a stale register read that only made the launch slower).
* `exec_prologue` -- round 4, the MaterialTransport N = 7 miscompute: at the top of the join block of a divergent `if`
                     (a branch target whose first exec write is the `s_or_b64 exec, exec, sN` of SI_END_CF) the register
                     allocator had placed its live-range split copies (`v_accvgpr_write aK, vJ`: VGPR -> AGPR saves) BEFORE
                     the exec restore, because an SGPR copy of the earlier SGPR allocation sat in front of it and ended
                     what LLVM takes for the block prologue.  The saves then run under the `then` branch's exec mask only;
                     lanes outside it later "restore" stale AGPR contents (a float multiplier in an integer sweep counter).
                     The check: no vector instruction between a branch-target label and the exec restore (`s_or_b64 exec, exec, sN` of
                     SI_END_CF, `s_or_saveexec_b64 sN, sN` of SI_ELSE) of a mask that was not saved inside that block.
* `resources`     -- VGPR / AGPR / scratch / LDS / occupancy from the code object's own metadata.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
VECTOR_PREFIXES = ("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")
BRANCHES = ("s_branch", "s_cbranch_", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    if not os.path.exists(p):
        p = shutil.which(name)
    if not p:
        raise RuntimeError(f"{name} not found (ROCm's LLVM expected under {LLVM_BIN})")
    return p


def extract_code_objects(so_path, workdir):
    """The gfx950 code objects bundled into a HIP shared library -> list of ELF paths (in workdir)."""
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([_tool("llvm-objdump"), "--offloading", os.path.basename(local)], cwd=workdir, check=True, capture_output=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn-amd-amdhsa--gfx950" in f and os.path.getsize(os.path.join(workdir, f)) > 0)


class Inst(object):
    __slots__ = ("addr", "op", "args", "target")

    def __init__(self, addr, op, args, target):
        self.addr, self.op, self.args, self.target = addr, op, args, target

    def __repr__(self):
        return f"{self.addr:#x}: {self.op} {self.args}"


_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_TGT = re.compile(r"<(\S+?)\+0x([0-9a-f]+)>\s*$")


def disassemble(code_object):
    """-> {kernel symbol: [Inst]} for the functions of one code object."""
    text = subprocess.run([_tool("llvm-objdump"), "-d", code_object], check=True, capture_output=True, text=True).stdout
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = _SYM.match(line)
        if m:
            base, cur = int(m.group(1), 16), []
            kernels[m.group(2)] = cur
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if op.startswith(("s_branch", "s_cbranch")):
            t = _TGT.search(line)
            if t:
                target = base + int(t.group(2), 16)
            elif line.rstrip().endswith(">"):   # a branch to the symbol itself (offset 0)
                target = base
        cur.append(Inst(addr, op, args, target))
    return kernels


def _vregs(text):
    regs = set()
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def dot_hazards(insts):
    """-> (number of DOT instructions, [problem strings])."""
    n, bad = 0, []
    for i, it in enumerate(insts):
        if not it.op.startswith("v_dot"):
            continue
        n += 1
        m = re.match(r"v(\d+)\b", it.args)
        if not m:
            bad.append(f"{it!r}: destination not understood")
            continue
        dst, waited, j = int(m.group(1)), 0, i + 1
        while waited < 3 and j < len(insts):
            nx = insts[j]
            if nx.op.startswith(BRANCHES):
                bad.append(f"control flow {waited} wait states after `{it!r}`")
                break
            if nx.op == "s_nop":
                waited += int(nx.args, 0) + 1
            else:
                if dst in _vregs(nx.args):
                    bad.append(f"`{nx!r}` touches v{dst} {waited} wait states after `{it!r}` (needs 3)")
                waited += 1
            j += 1
    return n, bad


def _sgpr_pair(text):
    m = re.search(r"\b(s\[\d+:\d+\]|vcc|s\d+)\b", text)
    return m.group(1) if m else None


def _is_then_entry(insts, index_of, label_addr, mask):
    """Is the block at label_addr entered ONLY straight from the saveexec that saved `mask` -- by the `s_cbranch_execnz` behind
    it (block placement moved the `then` body out of line) or by falling through the `s_cbranch_execz` behind it?  Then its
    instructions are MEANT to run under that mask, and the `s_or_b64 exec` that follows them is the body's own copy of the join."""
    def saves_mask_before(q):
        # `s_and_saveexec mask, cond` right behind the branch, or the same thing spelt out when the condition had to be fetched
        # first (round 5, an SGPR-spilled condition): `s_mov_b64 mask, exec ; v_readlane ... ; s_and_b64 sX, mask, sY ; s_mov_b64 exec, sX`
        for back in range(q - 1, max(q - 8, -1), -1):
            b = insts[back]
            if b.op.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec")) and b.args.split(",")[0].strip() == mask:
                return True
            if b.op == "s_mov_b64" and b.args.replace(" ", "") == f"{mask},exec".replace(" ", ""):
                return True
            if b.op.startswith(BRANCHES):
                return False
        return False

    k = index_of[label_addr]
    if k > 0 and not insts[k - 1].op.startswith(("s_branch", "s_endpgm", "s_setpc_b64")):   # reachable by falling through
        if not (insts[k - 1].op == "s_cbranch_execz" and saves_mask_before(k - 1)):
            return False
    preds = [q for q, it in enumerate(insts) if it.target == label_addr]
    return bool(preds) and all(insts[q].op == "s_cbranch_execnz" and saves_mask_before(q) for q in preds)


def _is_else_body(insts, index_of, label_addr, mask):
    """The block at `label_addr`, which ends in `s_or_b64 exec, exec, mask`, is the BODY of an `else` whose head was lowered as
    `s_or_saveexec_b64 mask, sX ; <copies for both sides> ; s_xor_b64 exec, exec, mask` (or as the single `s_andn2_saveexec_b64 mask, sX`):
    every way into it -- the fall-through and every branch -- comes straight from that instruction.  Its instructions are MEANT to run for the else lanes only
    (typically the else side's value of a phi, one v_mov) and the restore that follows is the join: not the defect."""
    k = index_of.get(label_addr)
    if k is None or k == 0:
        return False

    def is_xor(it):   # the second half of SI_ELSE: exec <- the else lanes, in either of the two forms the compiler emits
        a = it.args.replace(" ", "")
        return (it.op == "s_xor_b64" and a == f"exec,exec,{mask}".replace(" ", "")) or \
               (it.op == "s_andn2_saveexec_b64" and a.startswith(mask.replace(" ", "") + ","))
    ways = []
    if not insts[k - 1].op.startswith(("s_branch", "s_endpgm", "s_setpc")):
        ways.append(insts[k - 1])                        # fall-through
    for q, it in enumerate(insts):
        if it.target == label_addr:
            if it.op != "s_branch" or q == 0:
                return False
            ways.append(insts[q - 1])
    return bool(ways) and all(is_xor(w) for w in ways)


def exec_prologue(insts):
    """Vector instructions between a branch-target label and the `s_or_b64 exec, exec, sN` that restores a mask saved in
    another block (= the SI_END_CF of a divergent if / loop whose join block this is).  -> [problem strings]."""
    targets = {it.target for it in insts if it.target is not None}
    index_of = {it.addr: k for k, it in enumerate(insts)}
    bad = []
    for i, it in enumerate(insts):
        if it.op == "s_or_b64" and it.args.replace(" ", "").startswith("exec,exec,"):
            mask = it.args.split(",")[2].strip()          # SI_END_CF
        elif it.op == "s_or_saveexec_b64":
            mask = it.args.split(",")[1].strip()          # SI_ELSE: exec <- exec | mask at the head of the `else` block
        else:
            continue
        seen, j, at_label, nested = [], i, it.addr in targets, []
        while not at_label and j > 0:
            j -= 1
            p = insts[j]
            if p.op.startswith(BRANCHES):
                break                                   # fell through from a branch: not a join label
            dst = p.args.split(",")[0].strip() if p.args else ""
            is_save = p.op.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec"))
            if p.op == "s_or_b64" and p.args.replace(" ", "").startswith("exec,exec,"):
                nested.append(p.args.split(",")[2].strip())   # the end of an `if` that lies wholly inside this block
            elif is_save and nested and dst == nested[-1]:
                nested.pop()
            elif p.op.startswith("s_") and dst == mask and not nested:
                if p.op == "s_mov_b64" and p.args.split(",")[1].strip() != "exec":   # a copy of the saved mask: follow it
                    mask = p.args.split(",")[1].strip()
                else:
                    seen = None                         # the mask was saved in this block: an `if` that began here
                    break
            elif p.op.startswith(VECTOR_PREFIXES) and not nested and p.op not in ("v_readlane_b32", "v_writelane_b32"):
                seen.append(p)                          # (v_readlane / v_writelane ignore exec: SGPR spill slots, e.g. the mask itself)
            if p.addr in targets:
                at_label = True
        if at_label and seen and _is_then_entry(insts, index_of, insts[j].addr, mask):
            continue        # an out-of-line `then` body (entered only by `s_and_saveexec mask; s_cbranch_execnz`) ending in its own copy of the join
        if at_label and seen and it.op == "s_or_b64" and _is_else_body(insts, index_of, insts[j].addr, mask):
            continue        # the body of an `else` (entered only from `s_xor_b64 exec, exec, mask`) falling into the join
        if at_label and seen:
            bad.append(f"{len(seen)} vector instruction(s) run under the incoming exec mask before `{it!r}` restores it, e.g. `{seen[-1]!r}`")
    return bad


def resources(code_object):
    """-> {kernel symbol: {vgpr, agpr, sgpr, scratch, lds, spill, occupancy}} from the AMDGPU metadata note."""
    text = subprocess.run([_tool("llvm-readelf"), "--notes", code_object], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
        blk = ".agpr_count:" + blk

        def field(name, default=0):
            m = re.search(r"\.%s:\s+(\S+)" % re.escape(name), blk)
            return m.group(1) if m else default
        name = field("name", "")
        if not name:
            continue
        v = int(field("vgpr_count"))
        lds = int(field("group_segment_fixed_size"))
        granule = (v + 7) // 8 * 8
        occ = min(8, 512 // max(granule, 1))
        out[name] = {"vgpr": v, "agpr": int(field("agpr_count")), "sgpr": int(field("sgpr_count")), "scratch": int(field("private_segment_fixed_size")),
                     "lds": lds, "spill": int(field("vgpr_spill_count")), "occupancy": occ}
    return out


def scan_library(so_path):
    """-> {kernel: {"n_insts", "dots", "dot_hazards", "exec_prologue", "resources"}} over every code object of the library."""
    report = {}
    with tempfile.TemporaryDirectory() as d:
        for co in extract_code_objects(so_path, d):
            res = resources(co)
            for name, insts in disassemble(co).items():
                n, hz = dot_hazards(insts)
                report[name] = {"n_insts": len(insts), "dots": n, "dot_hazards": hz, "exec_prologue": exec_prologue(insts),
                                "resources": res.get(name, {})}
    return report


def show(so_path, kernel, addr, before=30, after=6):
    """Prints the instructions around `addr` of one kernel (to look at a finding)."""
    with tempfile.TemporaryDirectory() as d:
        for co in extract_code_objects(so_path, d):
            insts = disassemble(co).get(kernel)
            if not insts:
                continue
            targets = {it.target for it in insts if it.target is not None}
            k = next(i for i, it in enumerate(insts) if it.addr == addr)
            for it in insts[max(0, k - before):k + after]:
                print(("L " if it.addr in targets else "  ") + repr(it) + (f"   -> {it.target:#x}" if it.target is not None else ""))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    if len(sys.argv) > 1 and sys.argv[1] == "--show":   # --show lib.so kernel 0xaddr
        show(sys.argv[2], sys.argv[3], int(sys.argv[4], 16))
        sys.exit(0)
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "marbler_amd", "librobogym_hip.so")
    rep = scan_library(lib)
    nd = sum(r["dots"] for r in rep.values())
    bad = 0
    for name in sorted(rep):
        r = rep[name]
        for msg in r["dot_hazards"] + r["exec_prologue"]:
            bad += 1
            print(f"{name}: {msg}")
    print(f"{lib}: {len(rep)} functions, {sum(r['n_insts'] for r in rep.values())} instructions, {nd} DOT instructions, {bad} findings")
    sys.exit(1 if bad else 0)
