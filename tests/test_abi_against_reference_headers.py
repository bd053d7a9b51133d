"""include/ms2_plugin_abi.h restates the slice of mediastreamer2's plugin ABI the facades use (the real headers need
bctoolbox / oRTP, absent from the image, so -DMSMI355X_USE_REAL_MS2_HEADERS cannot be compiled here).  What CAN be
checked mechanically is checked here, in the container only: the reference headers under /root/reference/include are
PARSED (never copied) and every id, method macro, enum value and struct field list our header declares is compared
with them.  Skipped where the reference tree is absent (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include/mediastreamer2"
OURS = os.path.join(ROOT, "include", "ms2_plugin_abi.h")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers not present (GPU box)")


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    return re.sub(r"\\\n", " ", txt)  # continuation lines joined


def read_ref(*names):
    return "\n".join(strip_comments(open(os.path.join(REF, n)).read()) for n in names)


def ours():
    txt = strip_comments(open(OURS).read())
    return txt[txt.index("#else"):]  # the restated part, not the real-headers include list


def enum_values(txt, opener):
    """{name: value} of the enum whose text starts at `opener`, counting implicit values like the compiler"""
    m = re.search(opener + r"\s*\{(.*?)\}", txt, flags=re.S)
    assert m, opener
    out, nxt = {}, 0
    for item in m.group(1).split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            name, expr = [x.strip() for x in item.split("=", 1)]
            expr = re.sub(r"(\d+)u\b", r"\1", expr)
            nxt = int(eval(expr, {"__builtins__": {}}, dict(out)))  # noqa: S307 -- header constants: integers, +, <<
        else:
            name = item
        out[name] = nxt
        nxt += 1
    return out


def method_defines(txt):
    """{NAME: (macro, args...)} for every #define NAME MS_FILTER_*(...) line; whitespace-normalised arguments"""
    out = {}
    for m in re.finditer(r"#define\s+(\w+)\s+(MS_FILTER_(?:BASE_)?(?:METHOD|EVENT)(?:_NO_ARG)?)\s*\(([^)]*)\)", txt):
        args = tuple(re.sub(r"\s+", " ", a.strip()) for a in m.group(3).split(","))
        out[m.group(1)] = (m.group(2),) + args
    return out


def struct_fields(txt, opener):
    """the declarators of a struct body in order, normalised: 'type name' with pointer stars attached to the name"""
    m = re.search(opener + r"\s*\{", txt)
    assert m, opener
    i, depth, body = m.end(), 1, []
    while depth:
        c = txt[i]
        depth += (c == "{") - (c == "}")
        body.append(c)
        i += 1
    body = "".join(body[:-1])
    body = re.sub(r"#else.*?#endif", " ", body, flags=re.S)  # the 64-bit branch of a pointer-size #if (this target)
    body = re.sub(r"#[^\n]*", " ", body)
    body = re.sub(r"\{[^{}]*\}", "{}", body)  # nested union / struct bodies compared separately
    fields = []
    for decl in body.split(";"):
        decl = re.sub(r"\s+", " ", decl).strip()
        if decl:
            fields.append(re.sub(r"\(\s+", "(", re.sub(r"\s*\*\s*", " *", decl)))
    return fields


def test_filter_ids_match_allfilters_h():
    ref = enum_values(read_ref("allfilters.h"), r"typedef enum MSFilterId")
    mine = enum_values(ours(), r"typedef enum MSFilterId")
    assert len(mine) >= 20
    for name, val in mine.items():
        assert name in ref, name
        assert ref[name] == val, f"{name}: ms2_plugin_abi.h says {val}, allfilters.h counts {ref[name]}"


def test_interface_ids_match_msfilter_h():
    ref = enum_values(read_ref("msfilter.h"), r"enum _MSFilterInterfaceId")
    m = re.search(r"enum\s*\{\s*MSFilterInterfaceBegin.*?\}", ours(), flags=re.S)
    mine = enum_values(m.group(0), r"enum")
    for name, val in mine.items():
        assert ref[name] == val, (name, val, ref[name])


def test_category_flags_and_pixfmt_enums():
    ref = read_ref("msfilter.h", "msvideo.h")
    mine = ours()
    for opener in (r"enum _MSFilterCategory", r"enum _MSFilterFlags"):
        r, o = enum_values(ref, opener), enum_values(mine, opener)
        for name, val in o.items():
            assert r[name] == val, (name, val, r[name])
    r = enum_values(ref, r"typedef enum _MSPixFmt") if re.search(r"typedef enum _MSPixFmt", ref) else enum_values(
        ref[ref.index("MS_PIX_FMT_UNKNOWN") - 40:], r"typedef enum\s*\w*")
    o = enum_values(mine[mine.index("MS_PIX_FMT_UNKNOWN") - 40:], r"typedef enum")
    for name, val in o.items():
        assert r[name] == val, (name, val, r[name])
    ro, oo = enum_values(ref, r"typedef enum MSVideoOrientation") if "typedef enum MSVideoOrientation" in ref else None, None
    if ro:
        oo = enum_values(mine, r"typedef enum MSVideoOrientation")
        for name, val in oo.items():
            assert ro[name] == val


def test_method_and_event_macros_match():
    ref = method_defines(read_ref("msfilter.h", "msvolume.h", "msaudiomixer.h", "msequalizer.h", "msinterfaces.h", "flowcontrol.h",
                                  "mschanadapter.h", "msgenericplc.h", "msvideo.h"))
    mine = method_defines(ours())
    assert len(mine) >= 60, len(mine)
    missing = [n for n in mine if n not in ref]
    # macro helpers our header defines in terms of each other (not ids) are not in the reference under that shape
    assert not missing, missing
    for name, spec in mine.items():
        assert ref[name] == spec, f"{name}: ours {spec}, reference {ref[name]}"


def test_the_method_id_packing_macro():
    ref = read_ref("msfilter.h")
    pat = r"#define\s+MS_FILTER_METHOD_ID\s*\(_id_,\s*_cnt_,\s*_argsize_\)([^\n]*)\n"
    norm = lambda s: re.sub(r"\s+", "", s)  # noqa: E731
    a, b = re.search(pat, ref), re.search(pat, ours() + "\n")
    assert a and b
    assert norm(a.group(1)) == norm(b.group(1))


@pytest.mark.parametrize("header,opener", [
    ("msfilter.h", r"struct _MSFilterMethod"),
    ("msfilter.h", r"struct _MSFilterDesc"),
    ("msaudiomixer.h", r"typedef struct MSAudioMixerCtl"),
    ("msequalizer.h", r"typedef struct _MSEqualizerGain"),
    ("flowcontrol.h", r"typedef struct _MSAudioFlowControlDropEvent"),
    ("flowcontrol.h", r"typedef struct _MSAudioFlowControlConfig"),
    ("msvideo.h", r"typedef struct _MSPicture"),
    ("msvideo.h", r"struct _MSScalerDesc"),
    ("msqueue.h", r"typedef struct _MSCPoint"),
    ("msqueue.h", r"typedef struct _MSQueue"),
    ("msqueue.h", r"struct _MSBufferizer"),
])
def test_struct_field_lists_match(header, opener):
    a = struct_fields(read_ref(header), opener)
    b = struct_fields(ours(), opener)
    canon = lambda f: re.sub(r"\bstruct _MSFilter\b", "MSFilter", re.sub(r"\bstruct (_\w+)\b", r"struct \1", f))  # noqa: E731
    assert [canon(x) for x in a] == [canon(x) for x in b], (a, b)


def test_filter_and_ticker_layout_prefixes():
    """MSFilter up to `seen` and MSTicker up to `time`: the same fields in the same order (types modulo the ms_mutex_t =
    pthread_mutex_t / MSList = bctbx_list_t aliases of ortp/port.h, which our header spells out)."""
    names = lambda fields: [re.split(r"[ *]", f)[-1].split("[")[0] for f in fields]  # noqa: E731
    a = names(struct_fields(read_ref("msfilter.h"), r"struct _MSFilter"))
    b = names(struct_fields(ours(), r"typedef struct _MSFilter"))
    flat = lambda v: [y.strip() for x in v for y in x.split(",")]  # noqa: E731
    assert flat(a)[:len(flat(b))] == flat(b), (a, b)
    a = names(struct_fields(read_ref("msticker.h"), r"struct _MSTicker"))
    b = names(struct_fields(ours(), r"typedef struct _MSTicker"))
    upto = b.index("time") + 1
    assert a[:upto] == b[:upto], (a, b)
