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
            "smgpu_iter_begin", "smgpu_iter_mid", "smgpu_iter_end", "smgpu_set_layers"} <= set(used)
    for name, n in used.items():
        assert name in decl, f"{name} is not declared in include/smgpu.h"
        assert decl[name] == n, f"{name}: {n} arguments in the adapter, {decl[name]} in the header"
    # struct types and their fields
    structs = {m.group(2): m.group(1) for m in re.finditer(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", hdr, flags=re.S)}
    for var, typ in (("d", "smgpu_mesh_desc"), ("prm", "smgpu_params"), ("hd", "smgpu_halo_desc"), ("ld", "smgpu_layer_desc")):
        assert re.search(rf"\b{typ}\s+{var}\b", src), (typ, var)
        for f in set(re.findall(rf"\b{var}\.(\w+)\s*=", src)):
            assert re.search(rf"\b{f}\b", structs[typ]), f"{typ} has no field {f}"
    for macro in set(re.findall(r"\bSMGPU_[A-Z_]+\b", src)):
        assert re.search(rf"#define\s+{macro}\b|\b{macro}\s*=", hdr), macro


def test_wmake_files_follow_the_reference_layout():
    files = open(os.path.join(ROOT, "adapter", "Make", "files")).read()
    opts = open(os.path.join(ROOT, "adapter", "Make", "options")).read()
    assert "smoothMeshGPU.C" in files and "EXE =" in files
    for lib in ("-lfiniteVolume", "-lmeshTools", "-lsmgpu"):
        assert lib in opts
    assert "-I$(SMGPU_ROOT)/include" in opts
