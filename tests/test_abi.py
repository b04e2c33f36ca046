"""The C-ABI library loads and exports every symbol include/mvhdp.h declares.
No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mvhdp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mvhdp_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from mvtopicmodel_amd import _lib
    L = _lib.load_library()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_lib.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.mvhdp_version()


def test_create_rejects_bad_config_without_touching_a_device():
    import ctypes as C
    from mvtopicmodel_amd import _lib
    L = _lib.load_library()
    cfg = _lib.Config()
    cfg.num_topics = 0
    cfg.num_modalities = 1
    h = C.c_void_p()
    assert L.mvhdp_create(C.byref(cfg), C.byref(h)) == -1     # MVHDP_ERR_INVALID_ARG
    assert b"num_topics" in L.mvhdp_last_error(None)
    assert L.mvhdp_create(None, C.byref(h)) == -1
    assert L.mvhdp_destroy(None) == 0


def test_no_device_is_a_loud_error_not_a_fallback():
    """Without a gfx950 device the product refuses to run; it never routes to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mvtopicmodel_amd import NativeSampler, MvhdpError
    with pytest.raises(MvhdpError) as e:
        NativeSampler(10, [100])
    assert e.value.code == -4                                   # MVHDP_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """Nothing under mvtopicmodel_amd/ may import, load, link or call anything under oracle/."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle[/.]binding|oracle/|libmvhdp_oracle|\borc_[a-z_]+\s*\()")
    pkg = os.path.join(ROOT, "mvtopicmodel_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not pat.search(src), (dp, f, pat.search(src).group(0))


def test_activation_key_layout_is_defined_once():
    """include/mvhdp.h owns the layout of the activation key (MVHDP_ACT_*); the Python host side must agree with it
    (a binding written from the header -- JNI, ctypes -- decodes the multi-GPU winner with these shifts)."""
    from mvtopicmodel_amd import dist, native
    hdr = open(os.path.join(ROOT, "include", "mvhdp.h")).read()
    val = lambda name: int(re.search(r"#define\s+%s\s+(0x[0-9a-fA-F]+|\d+)" % name, hdr).group(1), 0)
    assert (val("MVHDP_ACT_DOC_SHIFT"), val("MVHDP_ACT_VIEW_SHIFT"), val("MVHDP_ACT_POS_SHIFT")) == \
           (dist.ACT_DOC_SHIFT, dist.ACT_VIEW_SHIFT, dist.ACT_POS_SHIFT) == (34, 31, 11)
    assert val("MVHDP_ACT_TOPIC_MASK") == dist.ACT_TOPIC_MASK == 0x7FF and val("MVHDP_ACT_VIEW_MASK") == dist.ACT_VIEW_MASK == 0x7
    # the fields do not overlap at the documented maxima: topic < 2048, position < 2^20, view < 8, entity < 2^29
    assert val("MVHDP_MAX_TOPICS") - 1 <= dist.ACT_TOPIC_MASK and (1 << 20) - 1 < (1 << (dist.ACT_VIEW_SHIFT - dist.ACT_POS_SHIFT))
    assert val("MVHDP_MAX_MODALITIES") - 1 <= dist.ACT_VIEW_MASK and dist.ACT_DOC_SHIFT + 29 <= 63
    key = (123456 << dist.ACT_DOC_SHIFT) | (3 << dist.ACT_VIEW_SHIFT) | (77 << dist.ACT_POS_SHIFT) | 1999
    assert dist.decode_activation(key) == (1999, 3) and dist.decode_activation(dist.KEY_NONE) == (-1, -1)
    # the sweep flags of the Python mirror are the header's
    for name in ("REUSE_TREES", "NO_APPLY", "EXACT_CHAIN", "GENERIC_KERNEL", "FROZEN", "LIVE", "SEGMENT_APPLY", "SEGMENT_OVERLAP"):
        assert getattr(native, "SWEEP_" + name) == int(re.search(r"#define\s+MVHDP_SWEEP_%s\s+(0x[0-9a-fA-F]+)u" % name, hdr).group(1), 16)
    assert native.SWEEP_LIVE_SEGMENTS(5) == 5 << 16


def test_one_hip_runtime_whatever_is_loaded_first():
    """The library first, torch second (the order that used to leave two HIP runtimes mapped): one
    libamdhip64 and one libhsa-runtime64 in the process, and torch.cuda still answers."""
    import subprocess
    import sys
    code = (
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from mvtopicmodel_amd import _lib\n"
        "_lib.load_library()\n"
        "import torch\n"
        "torch.cuda.is_available()\n"
        "m = _lib.mapped_runtime_libraries()\n"
        "assert len(m['libamdhip64']) == 1 and len(m['libhsa-runtime64']) <= 1, m\n"
        "print('one runtime', m['libamdhip64'][0])\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "one runtime" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_m0_is_touched_by_the_lane_write_sequences_only(tmp_path):
    """wave_writelane (csrc/mvhdp_wave.h) moves the lane select through M0 in inline assembly without saving it: that is sound only
    while the compiler itself never keeps a value in M0 in these kernels.  Disassemble the device code of the built library and check
    that every mention of m0 is `s_mov_b32 m0, sN` immediately followed by `v_writelane_b32 vN, sM, m0`."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [shutil.which("objcopy"), os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-objdump")]
    if not all(t and os.path.exists(t) for t in tools):
        pytest.skip("binutils / ROCm LLVM tools not available")
    lib = os.path.join(ROOT, "mvtopicmodel_amd", "lib", "libmvhdp.so")
    fat = tmp_path / "fat.bin"
    subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", lib, str(fat)], check=True)
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    assert starts, "no offload bundle in the library"
    seen = 0
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        blob, co = tmp_path / f"b{n}.bin", tmp_path / f"d{n}.co"
        blob.write_bytes(data[a:b])
        subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={blob}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        text = subprocess.run([tools[2], "-d", str(co)], check=True, capture_output=True, text=True).stdout
        ins = [l.split("//")[0].strip() for l in text.split("\n") if "//" in l]
        for i, l in enumerate(ins):
            if not re.search(r"\bm0\b", l):
                continue
            seen += 1
            if re.fullmatch(r"s_mov_b32 m0, s\d+", l):
                assert i + 1 < len(ins) and re.fullmatch(r"v_writelane_b32 v\d+, s\d+, m0", ins[i + 1]), (l, ins[i + 1])
            else:
                assert re.fullmatch(r"v_writelane_b32 v\d+, s\d+, m0", l) and re.fullmatch(r"s_mov_b32 m0, s\d+", ins[i - 1]), l
        # the hand-written row-broadcast scan steps (wave_incl_scan_f_dpp) carry their own s_nop for the VGPR-write -> DPP-read hazard;
        # the other DPP hazard -- a VALU write of EXEC needs five wait states before a DPP instruction -- is outside what the asm can
        # see: no v_cmpx may sit within the five instructions in front of one
        for i, l in enumerate(ins):
            if l.startswith("v_add_f32_dpp") and "row_bcast" in l:
                assert not any(x.startswith("v_cmpx") for x in ins[max(0, i - 6):i]), ins[max(0, i - 6):i + 1]
    assert seen > 0                                                          # (the sequences are there: the test looks at the right code)
