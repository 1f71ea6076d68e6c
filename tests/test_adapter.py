"""adapter/smoothMeshGPU.C cannot be compiled here (no OpenFOAM): at least every smgpu_* entry point / struct field it uses
must exist in include/smgpu.h with the same arity, so that the file does not rot silently."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", " ", t)


def _call_arity(text, pos):
    """number of top-level arguments of the call whose '(' is at text[pos]"""
    depth, n, seen = 0, 0, False
    for i in range(pos, len(text)):
        c = text[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif c == "," and depth == 1:
            n += 1
        elif depth >= 1 and not c.isspace():
            seen = True
    raise AssertionError("unbalanced call")


def test_adapter_uses_only_declared_entry_points():
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "smgpu.h")).read())
    src = _strip_comments(open(os.path.join(ROOT, "adapter", "smoothMeshGPU.C")).read())
    decl = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(smgpu_\w+)\s*\(", hdr):
        args = hdr[m.end() - 1:]
        n = _call_arity(args, 0)
        inner = args[1:args.index(")")].strip()
        decl[m.group(1)] = 0 if inner in ("void", "") else n
    used = {}
    for m in re.finditer(r"\b(smgpu_\w+)\s*\(", src):
        used.setdefault(m.group(1), _call_arity(src, m.end() - 1))
    assert {"smgpu_create", "smgpu_set_params", "smgpu_iterate", "smgpu_get_points", "smgpu_destroy", "smgpu_halo_configure",
            "smgpu_iter_begin", "smgpu_iter_mid", "smgpu_iter_end", "smgpu_set_layers", "smgpu_set_foam_variant",
            "smgpu_set_boundary_smoothing", "smgpu_boundary_begin", "smgpu_boundary_step", "smgpu_boundary_shared",
            "smgpu_layers_begin", "smgpu_layers_step", "smgpu_layers_shared", "smgpu_halo_l_doubles",
            "smgpu_get_boundary_classification"} <= set(used)
    for name, n in used.items():
        assert name in decl, f"{name} is not declared in include/smgpu.h"
        assert decl[name] == n, f"{name}: {n} arguments in the adapter, {decl[name]} in the header"
    # struct types and their fields
    structs = {m.group(2): m.group(1) for m in re.finditer(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", hdr, flags=re.S)}
    for var, typ in (("d", "smgpu_mesh_desc"), ("prm", "smgpu_params"), ("hd", "smgpu_halo_desc"), ("ld", "smgpu_layer_desc"),
                     ("bd", "smgpu_boundary_desc")):
        assert re.search(rf"\b{typ}\s+{var}\b", src), (typ, var)
        for f in set(re.findall(rf"\b{var}\.(\w+)\s*=", src)):
            assert re.search(rf"\b{f}\b", structs[typ]), f"{typ} has no field {f}"
    for macro in set(re.findall(r"\bSMGPU_[A-Z_]+\b", src)) - {"SMGPU_WITH_RCCL"}:      # (a build switch of the adapter itself)
        assert re.search(rf"#define\s+{macro}\b|\b{macro}\s*=", hdr), macro


def test_adapter_compiles_on_both_openfoam_lines_as_far_as_text_can_tell():
    """The reference builds against OpenFOAM.com v2312-v2506 AND OpenFOAM.org 12 (Allwmake:36-47) and therefore only uses the
    argList members both lines have (SM.C:1454-1458, 1788-1918).  The adapter must do the same, pick the engine's geometry
    variant from the define Allwmake passes, keep the reference's write rule and defaults, and hold an RCCL exchange on device
    pointers beside the Pstream one."""
    src = _strip_comments(open(os.path.join(ROOT, "adapter", "smoothMeshGPU.C")).read())
    # OpenFOAM.com-only members of argList (absent from OpenFOAM.org 12): must not appear
    for member in ("getOrDefault", "get<", "getList", "readIfPresent", "readListIfPresent", "found(", "lookup("):
        assert not re.search(r"\bargs\s*\.\s*" + re.escape(member), src), f"args.{member} exists on one OpenFOAM line only"
    # OpenFOAM.com-only spellings elsewhere
    for token in (".cdata()", "Foam::zero", "labelRange", "Pstream::broadcast"):
        assert token not in src, token
    # what the reference itself calls
    ref = open("/root/reference/src/smoothMesh.C").read() if os.path.exists("/root/reference/src/smoothMesh.C") else None
    for member in ("optionFound", "optionLookupOrDefault", "optionLookup", "optionRead", "setOption"):
        assert re.search(r"\bargs\s*\.\s*" + member + r"\b", src), member
        if ref is not None:
            assert re.search(r"\bargs\s*\.\s*" + member + r"\b", ref), member
    # geometry variant of the OpenFOAM line compiled against
    assert re.search(r"#if defined\(OPENFOAM_ORG\)\s+check\(smgpu_set_foam_variant\(h, SMGPU_FOAM_ORG\)\);", src)
    assert re.search(r"#elif defined\(OPENFOAM_COM\)[^\n]*\n\s+check\(smgpu_set_foam_variant\(h, SMGPU_FOAM_COM\)\);", src)
    assert "#error" in src
    # SM.C:1918 default and SM.C:2416 write rule
    assert 'optionLookupOrDefault("writeInterval", centroidalIters)' in src
    assert "((i + 1) % writeInterval) == 0) and (i > 0)" in src and "(i % writeInterval) == 0 && i > 1" in src
    # smoothMeshCommon.H:20
    assert "REL_TOL_ADAPTER = 1e-4" in src
    # the two transports
    assert "#ifdef SMGPU_WITH_RCCL" in src and "ncclSend(" in src and "ncclRecv(" in src and "ncclGroupStart()" in src
    assert "PstreamBuffers" in src
    # every option of the reference is declared (SM.C:1642-1784)
    if ref is not None:
        ref_opts = set(re.findall(r'argList::add(?:Bool)?Option\s*\(\s*"(\w+)"', ref))
        mine = set(re.findall(r'argList::add(?:Bool)?Option\s*\(\s*"(\w+)"', src))
        assert ref_opts <= mine, ref_opts - mine


def test_wmake_files_follow_the_reference_layout():
    files = open(os.path.join(ROOT, "adapter", "Make", "files")).read()
    assert "smoothMeshGPU.C" in files and "EXE =" in files
    for name, surf in (("options", "-lsurfMesh"), ("options.com", "-lsurfMesh"), ("options.org", "-ltriSurface")):
        opts = open(os.path.join(ROOT, "adapter", "Make", name)).read()
        for lib in ("-lfiniteVolume", "-lmeshTools", "-lsmgpu", surf, "$(SMGPU_COMM_LIBS)"):
            assert lib in opts, (name, lib)
        assert "-I$(SMGPU_ROOT)/include" in opts and "$(VERSION_SPECIFIC_INC)" in opts
    allw = open(os.path.join(ROOT, "adapter", "Allwmake")).read()
    assert "META-INFO" in allw and "-DOPENFOAM_COM" in allw and "-DOPENFOAM_ORG" in allw and "SMGPU_WITH_RCCL" in allw


import subprocess

import pytest


@pytest.mark.parametrize("defines", [["-DOPENFOAM_COM"], ["-DOPENFOAM_ORG"], ["-DOPENFOAM_COM", "-DSMGPU_WITH_RCCL"]])
def test_adapter_parses_and_type_checks_against_stand_in_headers(defines):
    """`g++ -fsyntax-only` of adapter/smoothMeshGPU.C against tests/adapter_stubs/ -- stand-in declarations written off the
    adapter's own call sites (TEST INFRASTRUCTURE: they are not OpenFOAM, pin nothing, and a wrong assumption about an OpenFOAM
    signature passes here).  What it does catch: plain C++ errors (round 4: `triSurface surf(fileName(x));` declared a function)
    and every use of include/smgpu.h -- argument types, struct fields, constness -- for both OpenFOAM lines and the RCCL build."""
    hip = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip, "hip", "hip_runtime.h")):
        pytest.skip("no ROCm headers")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=vexing-parse", "-D__HIP_PLATFORM_AMD__", *defines, "-I" + hip,
           "-I" + os.path.join(ROOT, "tests", "adapter_stubs"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "adapter", "smoothMeshGPU.C")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "warning" not in r.stderr, r.stderr[-4000:]


def test_adapter_must_name_its_openfoam_line():
    """without -DOPENFOAM_COM / -DOPENFOAM_ORG the adapter refuses to compile (#error): the two lines compute face centres differently"""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "tests", "adapter_stubs"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "adapter", "smoothMeshGPU.C")]
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("no ROCm headers")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "compile through adapter/Allwmake" in r.stderr


def test_stand_in_headers_say_what_they_are():
    d = os.path.join(ROOT, "tests", "adapter_stubs")
    for f in os.listdir(d):
        assert "TEST INFRASTRUCTURE" in open(os.path.join(d, f)).read(), f


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md's table 'Every entry point of include/smgpu.h' must name each declared function, and nothing that is
    not declared"""
    header = _strip_comments(open(os.path.join(ROOT, "include", "smgpu.h")).read())
    declared = set(re.findall(r"\b(smgpu_[a-z_0-9]+)\s*\(", header))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("## Every entry point of `include/smgpu.h`"):]
    named = set(re.findall(r"`(smgpu_[a-z_0-9]+)`", table))
    assert declared - named == set(), sorted(declared - named)
    assert named - declared == set(), sorted(named - declared)
