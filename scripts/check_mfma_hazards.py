#!/usr/bin/env python3
"""Static check of the shipped gfx950 code for the VALU-write -> MFMA-read issue distance.

gfx950 needs two wait states between a VALU instruction that writes a VGPR and a v_mfma that reads it as SrcA / SrcB.  hipcc keeps
that distance for its own instructions (s_nop included) with ONE exception found in round 4: a v_pk_fma_f32 one instruction in front
of the v_mfma_f32_32x32x2_f32 that reads its result (four sites in k_bwd_wave<0..2>: the matrix core read the previous step's
operand, the Qbar sums of those steps were wrong, visible in the R gradient once sigma^2 |R|^2 dt is not negligible -- DESIGN 4.3e).
The check therefore makes no exception for the compiler's own code.  hipcc also does not look inside inline asm: a VALU instruction written as asm
(or an asm v_mfma) can land closer, and the matrix core then reads the register's previous contents (round 4: an asm
v_cvt_pk_f16_f32 next to its v_mfma gave NaN gradients in k_fwd_rho_mfma on first launch only).  This script disassembles every
code object in libcmps.so and reports each v_mfma whose A / B operand was written by a VALU instruction fewer than two wait
states earlier, following fall-through and branch edges.

The same walk is done forwards for the second distance inline asm can break: a v_mfma's result may not be read (or overwritten) by
anything but a dependent v_mfma accumulating into it before the matrix core has written it -- passes + 3 wait states for the
4- and 8-pass bf16 / fp16 instructions (7 and 11), passes + 2 for the fp32 ones (the values hipcc itself keeps: the closest
compiler-placed readers in this library sit at 8, 12 and 18).

Third distance of the same kind (the wave reductions are asm DPP chains): a DPP instruction may not read, as its permuted operand,
a VGPR that a VALU instruction wrote fewer than two wait states earlier.

Round 5 added two hand-written sequences of the same kind to k_bwd_wave2w (cmps_wave_bwd2.hip), and two checks for them:
  * v_permlane32_swap / v_permlane16_swap may not read a VGPR that a VALU instruction wrote fewer than two wait states earlier (the
    chain step's three exchanges are asm with two v_pk_fma_f32 in those slots; hipcc pads its own with s_nop 1);
  * no instruction may read the destination of a ds_read / global_load / buffer_load before SOME s_waitcnt on its counter has been issued
    behind it (layout order).  The asm loads are awaited by waits in later statements, which hipcc's own s_waitcnt insertion does not
    know about: a copy of a loaded register that the compiler places in front of the wait -- pair re-packing in k_bwd_wave3's first form,
    loop-carried copies of the broadcast registers in k_fwd_wave2<no stash> -- reads whatever the register held before.

usage: check_mfma_hazards.py [libcmps.so]      exit code 1 when a violation is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NEED = 2                                                  # wait states between the VALU write and the MFMA read


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], capture_output=True)
    blob = open(fat, "rb").read() if os.path.exists(fat) else b""
    if not blob:                                          # a device-only object (hipcc --cuda-device-only -c): a bare bundle or code object
        blob = open(lib, "rb").read()
        if MAGIC not in blob:
            return [lib]
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for n, s in enumerate(starts):
        e = starts[n + 1] if n + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, f"bundle{n}.bin")
        open(part, "wb").write(blob[s:e])
        co = os.path.join(tmp, f"code{n}.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={part}", f"--output={co}"], check=True, capture_output=True)
        if os.path.getsize(co):
            out.append(co)
    return out


def vregs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def regs(tok):
    """(file, index) of every VGPR / AGPR a disassembled operand names"""
    tok = tok.strip().split(" ")[0]
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def result_wait_states(op):
    """wait states before a v_mfma's result may be read: passes + 3 (XDL), passes + 2 (fp32)"""
    m = re.match(r"v_mfma_f32_(\d+)x(\d+)x(\d+)_(\w+)", op)
    if not m:
        return 19
    mm, _, kk, ty = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    if ty == "f32":
        return {(32, 2): 16, (16, 4): 8, (32, 1): 16, (16, 1): 8, (4, 1): 2}.get((mm, kk), 16) + 2
    passes = {(16, 32): 4, (32, 16): 8, (16, 16): 4, (32, 8): 8}.get((mm, kk), 16)
    return passes + 3


def is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac", "v_accvgpr_write"))


def parse(dis):
    """functions: name -> list of (addr, opcode, operands)"""
    funcs, cur = {}, None
    for line in open(dis):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def check(funcs):
    bad, n_mfma = [], 0
    for name, ins in funcs.items():
        index = {a: i for i, (a, _, _) in enumerate(ins)}
        preds = {}                                        # instruction index -> indices of branches that jump to it
        # branch targets: the disassembler prints the word offset; the target address = addr + 4 + 4 * simm16
        for i, (a, op, args) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                t = a + 4 + 4 * int(args.split()[0])
                if t in index:
                    preds.setdefault(index[t], []).append(i)

        def walk(i, need, src, seen):
            """look back from instruction i (exclusive) while fewer than `need` wait states have passed"""
            j = i - 1
            while need > 0 and j >= 0:
                for p in preds.get(j + 1, []):            # a branch lands between j and j + 1
                    if (p, need) not in seen:
                        seen.add((p, need))
                        yield from walk(p + 1, need, src, seen)
                a, op, args = ins[j]
                if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    return                                # no fall-through from here
                if op == "s_nop":
                    need -= int(args.split()[0]) + 1
                    j -= 1
                    continue
                if is_valu(op) and args:
                    if vregs(args.split(",")[0]) & src:
                        yield j
                        return
                need -= 1
                j -= 1

        def walk_fwd(i, need, dst, seen):
            """instructions after i that touch dst before `need` wait states have passed (a v_mfma chaining on SrcC is allowed)"""
            j = i + 1
            while need > 0 and j < len(ins):
                a, op, args = ins[j]
                if op == "s_nop":
                    need -= int(args.split()[0]) + 1
                    j += 1
                    continue
                if op in ("s_endpgm", "s_setpc_b64"):
                    return
                if op.startswith(("s_cbranch", "s_branch")):
                    t = a + 4 + 4 * int(args.split()[0])
                    if t in index and (index[t], need) not in seen:
                        seen.add((index[t], need))
                        yield from walk_fwd(index[t] - 1, need - 1, dst, seen)
                    if op == "s_branch":
                        return
                    need -= 1
                    j += 1
                    continue
                toks = args.split(",") if args else []
                touched = set()
                if op.startswith(("v_mfma", "v_smfmac")):
                    touched = regs(toks[1]) | regs(toks[2])              # A / B read of a pending result
                    if not touched & dst and (regs(toks[0]) | regs(toks[3])) & dst:
                        return                                           # the next link of an accumulation chain: interlocked
                else:
                    for t in toks:
                        touched |= regs(t)
                if touched & dst:
                    yield j
                    return
                need -= 1
                j += 1

        def read_before_any_wait(i, dst, counter):
            """instructions behind i, in layout order, that READ dst before any s_waitcnt on `counter` (lgkmcnt / vmcnt) has been issued: a load
            whose result the compiler copied or used in front of the wait that makes it valid (inline-asm loads with the wait in a later
            statement: k_bwd_wave3's first form copied five such registers)"""
            for j in range(i + 1, min(i + 400, len(ins))):
                a, op, args = ins[j]
                if op == "s_waitcnt" and counter in args:
                    return
                if op in ("s_endpgm", "s_setpc_b64", "s_barrier") or op.startswith(("s_cbranch", "s_branch")) or j in preds:
                    return                                # straight-line code only: what another path does with the register is not this load's
                toks = args.split(",") if args else []
                srcs = toks[1:] if (op.startswith(("v_", "ds_read", "global_load", "buffer_load")) and not op.startswith("v_cmp")) else toks
                touched = set()
                for t in srcs:
                    touched |= regs(t)
                if touched & dst:
                    yield j
                    return
                if toks and op.startswith(("v_", "ds_read", "global_load")) and regs(toks[0]) & dst:
                    return                                # overwritten: the load's value is dead

        for i, (a, op, args) in enumerate(ins):
            if op.startswith(("v_permlane32_swap", "v_permlane16_swap")) and args:      # fourth distance: VALU write -> lane-exchange read
                toks = args.split(",")
                for j in walk(i, NEED, vregs(toks[0]) | vregs(toks[1]), set()):
                    bad.append((name, ins[j], ins[i]))
                continue
            if op.startswith(("global_load_dword", "buffer_load_dword")) and args:      # (asm loads included: hipcc does not count those)
                for j in read_before_any_wait(i, regs(args.split(",")[0]), "vmcnt"):
                    bad.append((name, ins[i], ins[j]))
                continue
            if op.startswith("ds_read") and args:
                for j in read_before_any_wait(i, regs(args.split(",")[0]), "lgkmcnt"):
                    bad.append((name, ins[i], ins[j]))
            if "_dpp" in op and args:                     # third distance: VALU write -> DPP read of the permuted operand (src0), 2 wait states
                toks = args.split(",")
                if len(toks) >= 2:
                    for j in walk(i, NEED, vregs(toks[1].strip().split(" ")[0]), set()):
                        bad.append((name, ins[j], ins[i]))
                continue
            if not op.startswith(("v_mfma", "v_smfmac")):
                continue
            n_mfma += 1
            ops = args.split(",")
            src = vregs(ops[1]) | vregs(ops[2])
            for j in walk(i, NEED, src, set()):
                bad.append((name, ins[j], ins[i]))
            for j in walk_fwd(i, result_wait_states(op), regs(ops[0]), set()):
                bad.append((name, ins[i], ins[j]))
    return bad, n_mfma


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "audio_mps_amd", "lib", "libcmps.so")
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(lib, tmp)
        for co in cos:
            dis = os.path.join(tmp, os.path.basename(co) + ".dis")
            with open(dis, "w") as f:
                subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, stdout=f)
            b, n = check(parse(dis))
            bad += b
            total += n
    for name, prod, cons in bad:
        print(f"HAZARD in {name}:\n    {prod[0]:08X}: {prod[1]} {prod[2]}\n    {cons[0]:08X}: {cons[1]} {cons[2]}")
    print(f"{len(cos)} code objects, {total} MFMA instructions checked, {len(bad)} closer than {NEED} wait states to a VALU producer or read before the result is written")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
