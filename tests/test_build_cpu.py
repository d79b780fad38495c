"""CPU: build-time audit of the 4 x 64 attention kernel (csrc/attn.hip::flash_fwd64_kernel).

That kernel OWNS accumulation registers a64..a255 (O^T and Q^T live there for a whole item, named literally inside asm
statements).  hipcc does not know: a spill, or any v_accvgpr_* it generates itself into that range, would corrupt them without
a fault or a message (MI355X guide 5.7 item 4).  So the device assembly is checked after every build: no scratch, no AGPR
touched outside an asm statement, the register file split leaves room for all 256 accumulation registers, and hipcc has put no
wait for memory (vmcnt) inside the tile loop other than the kernel's own counted ones - the LDS-DMA pieces are asm statements
precisely so that no s_waitcnt vmcnt(0) lands in front of the next LDS read."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def attn_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from g2vlm_amd import build
    out = tmp_path_factory.mktemp("asm") / "attn.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           *build.FILE_FLAGS.get("attn.hip", []), "-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC, "-S", "--cuda-device-only",
           os.path.join(build.CSRC, "attn.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


@pytest.mark.timeout(600)
def test_attn64_owns_its_accumulation_registers(attn_asm):
    s = attn_asm
    name = next(m for m in re.findall(r"^(_Z\S*flash_fwd64_kernel\S*):", s, re.M))
    i = s.index("\n", s.index(name + ":"))
    body = s[i:s.index("s_endpgm", i)]
    assert body.count("v_mfma_f32_32x32x16_bf16") >= 6 * 72
    assert "scratch_" not in body, "the kernel spills"
    in_asm, outside = False, []
    for ln in body.split("\n"):
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif not in_asm and not ln.strip().startswith(";") and ("accvgpr" in ln or re.search(r"[ ,\[]a\[?\d", ln)):
            outside.append(ln.strip())
    assert not outside, outside[:5]
    desc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", s, re.S).group(1)
    val = lambda k: int(re.search(r"\." + k + r"\s+(\d+)", desc).group(1))     # noqa: E731
    assert val("amdhsa_private_segment_fixed_size") == 0
    accum, nxt = val("amdhsa_accum_offset"), val("amdhsa_next_free_vgpr")
    assert nxt - accum == 256 and nxt <= 512, (accum, nxt)                        # a0..a255 allocated behind the arch VGPRs
    # inside the steady-state tile (the blocks that issue 8 DMA pieces) the only vmcnt waits are the kernel's own counted vmcnt(8)
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    steady = [b for b in blocks if b.count("global_load_lds_dwordx4") == 8 and b.count("v_mfma_f32_32x32x16_bf16") == 72]
    assert len(steady) == 2, len(steady)                                             # the two ping-pong copies
    for b in steady:
        waits = re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", b)
        assert waits == ["8"], waits
        n_instr = sum(1 for ln in b.split("\n") if ln.startswith("\t") and not ln.strip().startswith((";", ".")))
        assert n_instr <= 520, n_instr                                               # <= 7.2 instructions per MFMA (8 x 32 form: 10.4)
