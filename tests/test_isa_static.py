"""Static checks on the ISA of the shipped library (no GPU needed: the gfx950 code object inside libppocar.so is extracted and
disassembled with the ROCm LLVM tools, ~2 s).

1. The fp16 operand split of the default policy arithmetic (model.py:28-41 on the matrix cores; policy.hpp: split_pair_h) is
   COMPILER-GENERATED v_fma_mixlo_f16 / v_fma_mixhi_f16 -- three instructions per pair of values -- and not inline asm: the
   registers it writes are MFMA operands, and on gfx950 a VALU write needs two wait states before an MFMA reads the register.
   LLVM's hazard recogniser inserts them for its own instructions only.
2. A static hazard check over every kernel of the library: no v_mfma reads (A, B or C operand) a VGPR that a VALU instruction
   wrote fewer than two wait states earlier (an instruction issue = one wait state, s_nop N = N + 1).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "ppo-car_amd", "libppocar.so")

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(operand):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _split_operands(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse_kernels(dis_text):
    """{symbol: [(opcode, [operands]), ...]} from llvm-objdump -d output"""
    kernels, cur = {}, None
    for line in dis_text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        code = line.split("//")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        cur.append((parts[0], _split_operands(parts[1]) if len(parts) > 1 else []))
    return kernels


def mfma_valu_hazards(insts, need=2):
    """[(index, mfma, culprit)]: MFMAs that read a VGPR written by a VALU instruction fewer than `need` wait states earlier"""
    bad = []
    for i, (op, ops) in enumerate(insts):
        if not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
            continue
        reads = set()
        for o in ops[1:4]:
            reads |= _regs(o)
        waited, j = 0, i - 1
        while j >= 0 and waited < need:
            pop, pops = insts[j]
            if pop == "s_nop":
                waited += int(pops[0], 0) + 1
            else:
                is_valu = pop.startswith("v_") and not pop.startswith(("v_mfma", "v_smfmac")) and not pop.startswith("v_cmp") and pops
                if is_valu and (_regs(pops[0]) & reads):
                    bad.append((i, (op, ops), (pop, pops)))
                waited += 1
            j -= 1
    return bad


def test_the_hazard_checker_itself():
    text = "\n".join([
        "0000000000001000 <k>:",
        "\tv_fma_mixhi_f16 v16, s1, v12, v13 op_sel:[0,1,0] op_sel_hi:[0,1,0]   // 000000001000: D3A24010",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[10:13], v[16:19], v[0:3]          // 000000001008: D3D40000",
        "\tv_fma_mixhi_f16 v17, s1, v12, v13 op_sel:[0,1,0] op_sel_hi:[0,1,0]",
        "\ts_nop 0",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[10:13], v[16:19], v[0:3]",
        "\tv_fma_mixhi_f16 v18, s1, v12, v13 op_sel:[0,1,0] op_sel_hi:[0,1,0]",
        "\ts_nop 1",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[10:13], v[16:19], v[0:3]",
        "\tv_fma_mixhi_f16 v19, s1, v12, v13 op_sel:[0,1,0] op_sel_hi:[0,1,0]",
        "\tv_add_u32_e32 v40, 1, v41",
        "\tv_add_u32_e32 v42, 1, v41",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[10:13], v[16:19], v[0:3]",
    ])
    k = parse_kernels(text)["k"]
    bad = mfma_valu_hazards(k)
    assert [b[0] for b in bad] == [1, 4]          # zero and one wait state: hazards; s_nop 1 or two instructions: fine
    assert _regs("v[16:19]") == {("v", 16), ("v", 17), ("v", 18), ("v", 19)} and _regs("a[0:1]") == {("a", 0), ("a", 1)}


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not found under {LLVM}")
    if not os.path.exists(LIB):
        pytest.skip("libppocar.so is not built")
    d = tmp_path_factory.mktemp("isa")
    fat, co = str(d / "fat.bin"), str(d / "co.o")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", LIB, str(d / "unused.so")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}", "--unbundle"])
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
    ks = parse_kernels(dis)
    shutil.rmtree(d, ignore_errors=True)
    assert len(ks) > 100
    return ks


def test_fp16_operand_split_is_compiler_generated_mix_instructions(kernels):
    src = open(os.path.join(ROOT, "ppo-car_amd", "csrc", "kernels", "policy.hpp")).read()
    assert not re.search(r'asm[^;]*v_fma_mix', src), "the operand split must not be inline asm (outside the compiler's MFMA hazard model)"
    checked = 0
    for name, insts in kernels.items():
        prec2 = (re.match(r"_Z14rollout_kernelILi\d+ELi\d+ELi2E", name) or re.match(r"_Z20rollout_small_kernelILi\d+ELi\d+ELi2E", name)
                 or re.match(r"_Z13policy_kernelILi\d+ELb[01]ELi2E", name) or re.match(r"_Z20policy_pack16_kernelILi2E", name))
        if not prec2:
            continue
        ops = [op for op, _ in insts]
        lo, hi, back = ops.count("v_fma_mixlo_f16"), ops.count("v_fma_mixhi_f16"), sum(op.startswith("v_cvt_f32_f16") for op in ops)
        # (pairs of values: one mixlo + one mixhi; the 16-envs-per-workgroup small form also splits SINGLE observation entries where they
        # are produced -- rollout.hpp: write_pieces -- with a lone mixlo each)
        assert hi > 0 and lo >= hi, (name, lo, hi)
        assert back == 0, (name, "the split converts fp16 back to fp32: the five-instruction form")
        checked += 1
    assert checked >= 20


def test_no_mfma_reads_a_vgpr_a_valu_wrote_within_two_wait_states(kernels):
    n_mfma = 0
    for name, insts in kernels.items():
        n_mfma += sum(op.startswith("v_mfma") for op, _ in insts)
        bad = mfma_valu_hazards(insts)
        assert not bad, (name, bad[:3])
    assert n_mfma > 5000
