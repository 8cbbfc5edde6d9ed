"""Static conformance of the Clojure / JNA host (clj/src/raytrace_clj/gpu.clj, written without a JVM) and of INTEGRATION.md's Clojure
snippets against the C-ABI they bind (include/rtmi.h): every (call-int "rtmi_..." ...) form must name a declared symbol with the declared
number of arguments; every scalar argument must be coerced at the call site ((int ...) for int / int32_t / uint32_t, (long ...) for
int64_t / uint64_t, (double ...) for double: JNA marshals by the boxed Java type, so an un-coerced Long in an int32_t slot only works by
accident of the calling convention); every array argument must be a primitive array of the header's element type; handles are Pointers,
handle outputs PointerByReference, handle lists Pointer arrays.  No JVM is needed (or available): this is a reader over the source text."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtmi.h")
GPU_CLJ = os.path.join(ROOT, "clj", "src", "raytrace_clj", "gpu.clj")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")


# ---- the header: name -> list of parameter categories ------------------------------------------------------------------------------------
def _category(ctype):
    if re.search(r"\*\s*const\s*\*", ctype):  # T *const *: a read-only array of handles
        return "handle-array"
    t = re.sub(r"\s+", " ", ctype.replace("const ", " ").strip())
    t = t.replace(" *", "*").replace("* ", "*")
    table = {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "double": "f64", "char*": "string",
             "rtmi_ctx*": "handle", "rtmi_scene*": "handle", "rtmi_ctx**": "handle-out", "rtmi_scene**": "handle-out", "rtmi_scene**const": "handle-array",
             "rtmi_scene*const*": "handle-array", "int32_t*": "int[]", "double*": "double[]", "uint8_t*": "byte[]", "uint64_t*": "long[]", "int64_t*": "long[]",
             "void*": "device-pointer", "char*out": "string"}
    assert t in table, "unknown C type in rtmi.h: %r" % ctype
    return table[t]


def header_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b([a-z_0-9]+(?:\s+[a-z_0-9]+)*\s*\**)\s*\b(rtmi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(2), m.group(3).strip()
        cats = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)  # type, then the parameter name
                cats.append(_category(mm.group(1)))
        protos[name] = cats
    return protos


# ---- a reader for Clojure source: nested lists of tokens ---------------------------------------------------------------------------------
def read_forms(src):
    pos, n = 0, len(src)
    close = {"(": ")", "[": "]", "{": "}"}

    def skip():
        nonlocal pos
        while pos < n:
            c = src[pos]
            if c in " \t\r\n,":
                pos += 1
            elif c == ";":
                while pos < n and src[pos] != "\n":
                    pos += 1
            else:
                break

    def read():
        nonlocal pos
        skip()
        if pos >= n:
            return None
        c = src[pos]
        if c in close:
            pos += 1
            items = [c]
            while True:
                skip()
                assert pos < n, "unbalanced %s" % c
                if src[pos] == close[c]:
                    pos += 1
                    return items
                items.append(read())
        if c == '"':
            j = pos + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            tok = src[pos:j + 1]
            pos = j + 1
            return tok
        if c in "'`@~":  # reader macros that wrap the next form
            pos += 1
            return [c, read()]
        if c == "^":     # metadata / type hint: drop it, return the hinted form
            pos += 1
            read()
            return read()
        if c == "#":     # #( ... ) anonymous fn, #{ } set, #' var
            pos += 1
            if src[pos] == "'":
                pos += 1
            return ["#", read()]
        if c == "\\":    # character literal
            j = pos + 2
            while j < n and re.match(r"[A-Za-z0-9]", src[j]):
                j += 1
            tok = src[pos:j]
            pos = j
            return tok
        j = pos
        while j < n and src[j] not in " \t\r\n,()[]{}\";":
            j += 1
        tok = src[pos:j]
        pos = j
        return tok

    forms = []
    while True:
        f = read()
        if f is None:
            return forms
        forms.append(f)


def is_list(f, head=None):
    return isinstance(f, list) and len(f) > 1 and f[0] == "(" and (head is None or f[1] == head)


def walk(f):
    yield f
    if isinstance(f, list):
        for x in f[1:]:
            yield from walk(x)


def let_bindings(top):
    """symbol -> value form for every let / binding vector inside one top-level form (later bindings shadow earlier ones: fine for this file)"""
    env = {}
    for f in walk(top):
        if isinstance(f, list) and f[0] == "(" and len(f) > 2 and f[1] in ("let", "when-let", "if-let", "loop") and isinstance(f[2], list) and f[2][0] == "[":
            b = f[2][1:]
            for k in range(0, len(b) - 1, 2):
                if isinstance(b[k], str):
                    env[b[k]] = b[k + 1]
    return env


def map_values(form):
    """{:key value ...} literal -> dict"""
    out = {}
    items = form[1:]
    for k in range(0, len(items) - 1, 2):
        if isinstance(items[k], str) and items[k].startswith(":"):
            out[items[k]] = items[k + 1]
    return out


ARRAY_CTORS = {"int-array": "int[]", "double-array": "double[]", "byte-array": "byte[]", "long-array": "long[]"}


def classify(arg, env, flat_map, params):
    """category of one call-int argument form"""
    if isinstance(arg, str):
        if arg.startswith('"'):
            return "string"
        if re.match(r"^-?[0-9]", arg):
            return "bare-number"
        if arg in env:
            return classify(env[arg], env, flat_map, params)
        if arg in params:
            return "param"
        return "unknown-symbol:" + arg
    if is_list(arg):
        head = arg[1]
        if head in ("int", "long", "double"):
            return {"int": "i32", "long": "i64", "double": "f64"}[head]
        if head in ARRAY_CTORS:
            return ARRAY_CTORS[head]
        if head == "PointerByReference.":
            return "handle-out"
        if head in (".getValue", "create-scene!"):
            return "handle"
        if head == "into-array":
            return "handle-array" if arg[2] in ("com.sun.jna.Pointer", "Pointer") else "array-of-" + str(arg[2])
        if head in ("first", "nth", "second"):  # an element of a vector of handles
            inner = classify(arg[2], env, flat_map, params)
            return "handle" if inner in ("handles", "param") else "element-of-" + inner
        if head == "mapv":  # (mapv (fn [...] ... (.getValue x)) coll): a vector of handles
            return "handles" if any(is_list(x, ".getValue") for x in walk(arg)) else "vector"
        if isinstance(head, str) and head.startswith(":") and len(arg) == 3 and isinstance(arg[2], str):  # (:key f): a field of flatten-scene's result
            v = flat_map.get(head)
            assert v is not None, "flatten-scene returns no %s" % head
            c = classify(v, {}, flat_map, set())
            return {"i32": "boxed-int"}.get(c, c)
    return "unclassified:" + repr(arg)[:60]


COMPATIBLE = {
    "i32": {"i32"}, "i64": {"i64"}, "f64": {"f64"}, "string": {"string"},
    "handle": {"handle", "param"},  # a function / doseq parameter that can only be a handle where the header wants one
    "handle-out": {"handle-out"}, "handle-array": {"handle-array"},
    "int[]": {"int[]"}, "double[]": {"double[]"}, "byte[]": {"byte[]"}, "long[]": {"long[]"},
}


def check_calls(forms, protos, flat_map, strict_arrays, where):
    n_calls = 0
    for top in forms:
        env = let_bindings(top)
        params = set()
        for f in walk(top):  # parameters of defn / fn / doseq / for in this top-level form
            if isinstance(f, list) and f[0] == "(" and len(f) > 2 and f[1] in ("fn", "defn", "defn-", "doseq", "for"):
                for v in f[2:5]:
                    if isinstance(v, list) and v[0] == "[":
                        params.update(x for x in v[1:] if isinstance(x, str))
        for f in walk(top):
            if not is_list(f, "call-int") or not (isinstance(f[2], str) and f[2].startswith('"rtmi_')):
                continue
            name, args = f[2].strip('"'), f[3:]
            n_calls += 1
            assert name in protos, "%s: %s is not declared in include/rtmi.h" % (where, name)
            want = protos[name]
            assert len(args) == len(want), "%s: %s takes %d arguments, the call passes %d" % (where, name, len(want), len(args))
            for k, (a, w) in enumerate(zip(args, want)):
                got = classify(a, env, flat_map, params)
                if not strict_arrays and w not in ("i32", "i64", "f64") and (got.startswith("unknown-symbol") or got == "param"):
                    continue  # documentation snippet: arrays / handles are named, not constructed
                assert got in COMPATIBLE[w], "%s: %s argument %d: header wants %s, the call passes %s (%r)" % (where, name, k + 1, w, got, a)
    return n_calls


def test_header_prototypes_are_all_parsed():
    import raytrace_clj_amd._ffi as ffi
    protos = header_prototypes()
    assert sorted(protos) == sorted(ffi.SYMBOLS), "the reader must see every prototype of include/rtmi.h"
    assert protos["rtmi_render"] == ["handle", "i32", "i32", "i32", "i32", "i64", "i32", "i32", "i32", "i32", "i32", "double[]", "byte[]", "long[]"]
    assert protos["rtmi_scene_create_ex"][-6:] == ["int[]", "int[]", "i32", "int[]", "double[]", "handle-out"] and len(protos["rtmi_scene_create_ex"]) == 21
    assert protos["rtmi_render_multi"][:2] == ["i32", "handle-array"]


def test_gpu_clj_calls_conform_to_the_header():
    protos = header_prototypes()
    forms = read_forms(open(GPU_CLJ).read())
    flat = [f for f in forms if is_list(f, "defn") and f[2] == "flatten-scene"]
    assert len(flat) == 1
    maps = [f for f in walk(flat[0]) if isinstance(f, list) and f[0] == "{" and any(x == ":prim-kind" for x in f[1:])]
    assert len(maps) == 1, "flatten-scene's result map"
    flat_map = map_values(maps[0])
    n = check_calls(forms, protos, flat_map, True, "gpu.clj")
    assert n >= 10
    called = {f[2].strip('"') for top in forms for f in walk(top) if is_list(f, "call-int")}
    # the one-GPU path and the multi-GPU path must both create the scene WITH its Perlin tables, ImageMap pixels and media calls
    for need in ("rtmi_init", "rtmi_scene_create_ex", "rtmi_scene_set_perlin", "rtmi_scene_set_images", "rtmi_scene_set_media_calls", "rtmi_scene_set_media_mode", "rtmi_render",
                 "rtmi_scene_clone", "rtmi_render_multi", "rtmi_scene_destroy", "rtmi_shutdown"):
        assert need in called, need
    by_name = {f[2]: f for f in forms if isinstance(f, list) and len(f) > 2 and f[1] in ("defn", "defn-")}
    for entry in ("render", "render-multi"):
        uses = {x[1] for x in walk(by_name[entry]) if is_list(x)}
        assert "create-scene!" in uses, "%s must build its scene through create-scene! (images, Perlin tables, media calls)" % entry
    helper = {f[2].strip('"') for f in walk(by_name["create-scene!"]) if is_list(f, "call-int")}
    assert helper == {"rtmi_scene_create_ex", "rtmi_scene_set_perlin", "rtmi_scene_set_images", "rtmi_scene_set_media_calls", "rtmi_scene_set_media_mode", "rtmi_scene_set_media_calls_narrowed"}


def test_integration_md_snippets_conform_to_the_header():
    protos = header_prototypes()
    text = open(INTEGRATION).read()
    blocks = re.findall(r"```clojure\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 3
    n = 0
    for b in blocks:
        forms = read_forms(b.replace("…", " "))
        n += check_calls(forms, protos, {}, False, "INTEGRATION.md")
    assert n >= 3


def test_reader_rejects_what_it_should():
    """the checker itself: an un-coerced scalar, a wrong array type, a wrong arity and an undeclared symbol are all caught"""
    import pytest
    protos = header_prototypes()
    flat_map = {":cam-kind": ["(", "int", ["(", ":kind", "cam"]], ":cam": ["(", "double-array", "x"], ":prim-kind": ["(", "int-array", "x"]}
    bad = [
        '(defn f [scn] (let [lin (double-array 3) rgb (byte-array 3) cnt (long-array 2)] (call-int "rtmi_render" scn (int 1) (int 1) (int 1) (int 1) 7 (int 0) (int 0) (int 0) (int 1) (int 1) lin rgb cnt)))',
        '(defn f [scn] (let [lin (double-array 3) rgb (byte-array 3) cnt (int-array 2)] (call-int "rtmi_render" scn (int 1) (int 1) (int 1) (int 1) (long 7) (int 0) (int 0) (int 0) (int 1) (int 1) lin rgb cnt)))',
        '(defn f [scn] (call-int "rtmi_scene_destroy" scn (int 0)))',
        '(defn f [scn] (call-int "rtmi_scene_frobnicate" scn))',
        '(defn g [ctx f] (let [scn (PointerByReference.)] (call-int "rtmi_scene_create" ctx (int 1) (:prim-kind f) (:cam f) (:prim-kind f) (int 1) (:prim-kind f) (:prim-kind f) (:cam f) (int 1) (:prim-kind f) (:cam f) (:prim-kind f) (:cam-kind f) (:cam f) scn)))',
    ]
    for src in bad:
        with pytest.raises(AssertionError):
            check_calls(read_forms(src), protos, flat_map, True, "synthetic")
    good = '(defn g [ctx f] (let [scn (PointerByReference.)] (call-int "rtmi_scene_create" ctx (int 1) (:prim-kind f) (:cam f) (:prim-kind f) (int 1) (:prim-kind f) (:prim-kind f) (:cam f) (int 1) (:prim-kind f) (:cam f) (:prim-kind f) (int (:cam-kind f)) (:cam f) scn)))'
    assert check_calls(read_forms(good), protos, flat_map, True, "synthetic") == 1


# ---- the flattener reads the reference's records by field keyword: every (:field record) must name a field the record has ------------------------------
# the reference's record declarations (defrecord Name [fields]) -- hitable.clj:15,36,97,141,180,224,269,301,333,375,391,410,491,516,548;
# shader.clj:29,46,76,114,129; texture.clj:14,26,44,60,74,88,103,113,126; camera.clj:8,35 -- as data
REFERENCE_RECORDS = {
    "Hitlist": ["items"], "bvh_node": ["left", "right", "box"], "UVSphere": ["center", "radius", "material"], "Sphere": ["center", "radius", "material"],
    "MovingSphere": ["center0", "t0", "center1", "t1", "radius", "material"], "RectXY": ["x0", "y0", "x1", "y1", "k", "material"],
    "RectXZ": ["x0", "z0", "x1", "z1", "k", "material"], "RectYZ": ["y0", "z0", "y1", "z1", "k", "material"], "FlipNormals": ["item"],
    "Translate": ["item", "offset"], "RotateY": ["obj", "rotated-bbox", "sin-theta", "cos-theta"], "Box": ["p0", "p1", "sides"],
    "ConstantMedium": ["boundary", "density", "phase-fn"], "Triangle": ["v0", "v1", "v2", "material"],
    "Lambertian": ["albedo"], "Metal": ["albedo", "fuzz"], "Dielectric": ["ri"], "DiffuseLight": ["tex"], "Isotropic": ["albedo"],
    "Constant": ["color"], "UVGradient": ["co", "cu", "cv", "cuv"], "Checkerboard": ["tex0", "tex1", "scale"], "PerlinNoise": ["scale"],
    "PerlinTurbulence": ["scale", "depth"], "Marble": ["scale", "depth"], "FlipTextureU": ["tex"], "FlipTextureV": ["tex"], "ImageMap": ["image"],
    "PinholeCamera": ["origin", "lleft", "horiz", "vert"], "ThinLensCamera": ["origin", "lleft", "horiz", "vert", "u", "v", "w", "aperture", "t0", "t1"],
}


def _field_reads(form, subject):
    """keywords read off `subject` inside form: (:kw subject)"""
    return [f[1][1:] for f in walk(form) if is_list(f) and len(f) == 3 and isinstance(f[1], str) and f[1].startswith(":") and f[2] == subject]


def test_gpu_clj_reads_only_fields_the_reference_records_have():
    forms = read_forms(open(GPU_CLJ).read())
    checked = 0
    seen_types = set()
    for top in forms:
        for f in walk(top):
            if is_list(f, "extend-protocol"):  # TypeName (method [this ...] body) ...
                items, k = f[3:], 0
                while k < len(items):
                    t = items[k]
                    k += 1
                    while k < len(items) and isinstance(items[k], list):
                        impl = items[k]
                        k += 1
                        if t == "Object":
                            continue
                        assert t in REFERENCE_RECORDS, "gpu.clj extends a record the reference does not declare: %s" % t
                        seen_types.add(t)
                        this = impl[2][1]  # first parameter of (leaves [this c f l b] ...)
                        for kw in _field_reads(impl, this):
                            assert kw in REFERENCE_RECORDS[t], "(:%s %s) -- %s has fields %s" % (kw, t, t, REFERENCE_RECORDS[t])
                            checked += 1
            if is_list(f, "condp") and len(f) > 4 and f[2] == "instance?":  # (condp instance? x Type expr Type expr ... [default])
                subject, clauses = f[3], f[4:]
                for j in range(0, len(clauses) - 1, 2):
                    t, expr = clauses[j], clauses[j + 1]
                    if not isinstance(t, str):
                        continue
                    assert t in REFERENCE_RECORDS, "gpu.clj dispatches on a record the reference does not declare: %s" % t
                    seen_types.add(t)
                    for kw in _field_reads(expr, subject):
                        assert kw in REFERENCE_RECORDS[t], "(:%s %s) -- %s has fields %s" % (kw, t, t, REFERENCE_RECORDS[t])
                        checked += 1
    assert checked >= 60, checked
    # every record the GPU path claims is handled somewhere in the flattener
    assert seen_types >= set(REFERENCE_RECORDS) - {"AABB"}, set(REFERENCE_RECORDS) - seen_types
    # and the imports name exactly those classes
    imports = [f for f in walk(forms[0]) if is_list(f, ":import")]
    imported = {x for f in imports for grp in f[2:] if isinstance(grp, list) for x in grp[2:] if isinstance(x, str)}
    assert set(REFERENCE_RECORDS) <= imported, set(REFERENCE_RECORDS) - imported


# ---- clj/test/raytrace_clj/gpu_test.clj: the test a maintainer runs first (never run here: no JVM) -- read statically ----------------------------------
# the reference's scene functions and their parameter lists (scene.clj:9, 230, 318), as data
REFERENCE_SCENE_FNS = {"make-two-spheres": 2, "make-cornell-box": 2, "make-random-scene": 4}
GPU_TEST_CLJ = os.path.join(ROOT, "clj", "test", "raytrace_clj", "gpu_test.clj")
CLJ_FIXTURES = os.path.join(ROOT, "clj", "test", "resources")


def _edn_keys(text):
    """top-level keywords of the fixture map (the files are written by scripts/export_clj_fixtures.py: one `:key value` per line)"""
    import re
    assert text.count("{") == text.count("}") == 1 and text.count("[") == text.count("]")
    return set(re.findall(r"(?m)^[ {]:([a-z0-9-]+) ", text)) | set(re.findall(r" :(ny|nx) ", text))


def test_gpu_test_clj_refers_to_things_that_exist():
    """the Clojure test calls the reference's scene functions with the arity scene.clj declares, gpu.clj vars that gpu.clj defines, reads fixture keys the
    exported fixtures hold, and every fixture is the scene of the golden .npz it was exported from (the exporter asserts that when it runs)"""
    forms = read_forms(open(GPU_TEST_CLJ).read())
    ns = forms[0]
    assert is_list(ns, "ns") and ns[2] == "raytrace-clj.gpu-test"
    requires = {v[1]: v[3] for req in walk(ns) if is_list(req, ":require") for v in req[2:] if isinstance(v, list) and v[0] == "[" and len(v) >= 4 and v[2] == ":as"}
    assert requires.get("raytrace-clj.scene") == "scene" and requires.get("raytrace-clj.gpu") == "gpu" and requires.get("clojure.edn") == "edn"
    gpu_defs = {f[2] for f in read_forms(open(GPU_CLJ).read()) if is_list(f) and f[1] in ("defn", "def") and isinstance(f[2], str)}
    calls = {"scene": [], "gpu": []}
    for top in forms:
        for f in walk(top):
            if is_list(f) and isinstance(f[1], str) and "/" in f[1]:
                alias, name = f[1].split("/", 1)
                if alias in calls:
                    calls[alias].append((name, len(f) - 2))
    assert {n for n, _ in calls["scene"]} == set(REFERENCE_SCENE_FNS)
    for name, argc in calls["scene"]:
        assert argc == REFERENCE_SCENE_FNS[name], "(scene/%s ...) with %d arguments, scene.clj declares %d" % (name, argc, REFERENCE_SCENE_FNS[name])
    assert {n for n, _ in calls["gpu"]} <= gpu_defs and {"flatten-scene", "render"} <= {n for n, _ in calls["gpu"]}
    # fixtures: present, balanced, and holding every key the test reads off them
    read_keys = {f[1][1:] for top in forms for f in walk(top) if is_list(f) and len(f) == 3 and isinstance(f[1], str) and f[1].startswith(":") and f[2] in ("fx", ["~", "fx"])}
    names = {f[2].strip('"') for top in forms for f in walk(top) if is_list(f, "fixture") and len(f) == 3 and isinstance(f[2], str) and f[2].startswith('"')}
    assert names == {"two_spheres", "cornell_box", "cover_n3"} and read_keys >= {"scene-seed", "bvh-axes", "nx", "ny", "cam", "prim-kind"}
    for n in names:
        keys = _edn_keys(open(os.path.join(CLJ_FIXTURES, n + ".edn")).read())
        assert read_keys <= keys | {"prim-flip", "prim-xform", "xform-kind", "xform-param"}, (n, read_keys - keys)
        assert {"scene-seed", "bvh-axes", "prim-kind", "prim-geom", "cam", "cam-kind"} <= keys
    # ... and the keys it compares are keys flatten-scene's result map has
    flat = [f for f in read_forms(open(GPU_CLJ).read()) if is_list(f, "defn") and f[2] == "flatten-scene"][0]
    flat_keys = {m[k][1:] for m in walk(flat) if isinstance(m, list) and m[0] == "{" for k in range(1, len(m), 2) if isinstance(m[k], str) and m[k].startswith(":")}
    compared = {f[2][1:] for top in forms for f in walk(top) if is_list(f, "get") and len(f) == 4 and f[2] == "f" or False for _ in [0]} if False else set()
    for top in forms:
        for f in walk(top):
            if isinstance(f, list) and f[0] == "[" and len(f) == 3 and all(isinstance(x, str) and x.startswith(":") for x in f[1:]):
                compared.add(f[1][1:])
            if is_list(f) and len(f) == 3 and isinstance(f[1], str) and f[1].startswith(":") and f[2] in ("f", "f1", "f2", "out"):
                compared.add(f[1][1:])
    assert compared - {"rgb8", "total-pixels", "total-rays"} <= flat_keys, compared - flat_keys
    assert {"prim-kind", "prim-geom", "cam", "cam-kind", "n-prims", "xform-param"} <= compared


def test_clj_fixtures_are_the_golden_scenes():
    """clj/test/resources/*.edn are re-exported from tests/golden/*.npz by the committed script and must not drift from them"""
    import subprocess
    import sys
    before = {n: open(os.path.join(CLJ_FIXTURES, n)).read() for n in sorted(os.listdir(CLJ_FIXTURES))}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "export_clj_fixtures.py")], stdout=subprocess.DEVNULL)
    after = {n: open(os.path.join(CLJ_FIXTURES, n)).read() for n in sorted(os.listdir(CLJ_FIXTURES))}
    assert before == after
