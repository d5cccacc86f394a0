#!/usr/bin/env python3
"""Extract the reference's golden vectors for the index-map path into JSON fixtures.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_golden.py

Sources (all in /root/reference/tests/ctest/api_tests.cc):
  * kExpected{Default,ColumnMajor,GdimsDist}PencilInfo   (:92-153)  -> pencil_info.json
  * expectShiftedRanks(...) tables, both rank orders      (:1380-1408) -> shifted_rank.json
  * dtype sizes (:449-459), backend strings (:467-493), defaults (:254-317) -> api_constants.json

Only DATA is extracted (numbers / strings the reference's tests assert); no reference source text
is stored.  The fixtures travel to the GPU box; /root/reference does not.
"""
import json
import os
import re
import sys

REF = "/root/reference/tests/ctest/api_tests.cc"
OUT = os.path.dirname(os.path.abspath(__file__))


def ints(s):
    return [int(x) for x in re.findall(r"-?\d+", s)]


def parse_constants(src):
    c = {}
    for name in ("kGdims", "kGdimsDist", "kPdims", "kHaloExtents", "kPadding"):
        m = re.search(r"constexpr std::array<int32_t, \d> %s\{([^}]*)\}" % name, src)
        c[name] = ints(m.group(1))
    m = re.search(r"constexpr std::array<bool, 3> kHaloPeriods\{([^}]*)\}", src)
    c["kHaloPeriods"] = [x.strip() == "true" for x in m.group(1).split(",")]
    return c


def parse_pencil_table(src, name, consts):
    m = re.search(r"%s\[3\]\[kApiTestRanks\] = \{(.*?)\n\};" % name, src, re.S)
    body = m.group(1)
    rows = re.findall(
        r"\{\{([^}]*)\}, \{([^}]*)\}, \{([^}]*)\}, \{([^}]*)\}, kHaloExtents, kPadding, (\d+)\}", body)
    assert len(rows) == 12, (name, len(rows))
    table = []
    for k, (shape, lo, hi, order, size) in enumerate(rows):
        table.append({
            "axis": k // 4, "rank": k % 4,
            "shape": ints(shape), "lo": ints(lo), "hi": ints(hi), "order": ints(order),
            "halo_extents": consts["kHaloExtents"], "padding": consts["kPadding"], "size": int(size),
        })
    return table


def parse_shifted(src, test_name):
    m = re.search(r"TEST_F\(ApiGetShiftedRankTest, %s\) \{(.*?)\n\}" % test_name, src, re.S)
    rows = re.findall(
        r"expectShiftedRanks\(active_comm_, handle_, grid_desc, (\d), (\d), (-?\d), (true|false), \{([^}]*)\}\);",
        m.group(1))
    return [{"axis": int(a), "dim": int(d), "displacement": int(s), "periodic": p == "true",
             "expected_by_rank": ints(e)} for a, d, s, p, e in rows]


def parse_strings(src, fn):
    return {name: text for text, name in re.findall(r'EXPECT_STREQ\("([^"]*)", %s\((CUDECOMP_\w+)\)\);' % fn, src)}


def main():
    if not os.path.exists(REF):
        sys.exit("reference not mounted; fixtures are already committed")
    src = open(REF).read()
    consts = parse_constants(src)

    pencil = {
        "source": "tests/ctest/api_tests.cc:72-153,1248-1290",
        "gdims": consts["kGdims"], "pdims": consts["kPdims"], "gdims_dist_case": consts["kGdimsDist"],
        "halo_extents": consts["kHaloExtents"], "padding": consts["kPadding"],
        "row_major": parse_pencil_table(src, "kExpectedDefaultPencilInfo", consts),
        "col_major": parse_pencil_table(src, "kExpectedColumnMajorPencilInfo", consts),
        "gdims_dist": parse_pencil_table(src, "kExpectedGdimsDistPencilInfo", consts),
    }
    json.dump(pencil, open(os.path.join(OUT, "pencil_info.json"), "w"), indent=1)

    shifted = {
        "source": "tests/ctest/api_tests.cc:1380-1408",
        "gdims": consts["kGdims"], "pdims": consts["kPdims"],
        "row_major": parse_shifted(src, "ReturnsExpectedRanksForRowMajorLayout"),
        "col_major": parse_shifted(src, "ReturnsExpectedRanksForColumnMajorLayout"),
    }
    assert len(shifted["row_major"]) == 6 and len(shifted["col_major"]) == 6
    json.dump(shifted, open(os.path.join(OUT, "shifted_rank.json"), "w"), indent=1)

    sizes = {name: int(v) for name, v in re.findall(
        r"cudecompGetDataTypeSize\((CUDECOMP_\w+), &dtype_size\)\);\s*EXPECT_EQ\((\d+), dtype_size\);", src)}
    m = re.search(r"TEST\(ApiGridDescAutotuneOptionsSetDefaultsTest, SetsDocumentedDefaults\) \{(.*?)\n\}", src, re.S)
    opt_defaults = {k: v for v, k in re.findall(r"EXPECT_EQ\(([\w.]+), options\.(\w+)\);", m.group(1))}
    opt_defaults.update({k: "true" for k in re.findall(r"EXPECT_TRUE\(options\.(\w+)\);", m.group(1))})
    opt_defaults.update({k: "false" for k in re.findall(r"EXPECT_FALSE\(options\.(\w+)\);", m.group(1))})
    m = re.search(r"TEST\(ApiGridDescConfigSetDefaultsTest, SetsDocumentedDefaults\) \{(.*?)\n\}", src, re.S)
    cfg_defaults = {k: v for v, k in re.findall(r"EXPECT_EQ\(([\w.]+), config\.(\w+)\);", m.group(1))}
    api = {
        "source": "tests/ctest/api_tests.cc:254-317,449-493",
        "dtype_sizes": sizes,
        "transpose_backend_strings": parse_strings(src, "cudecompTransposeCommBackendToString"),
        "halo_backend_strings": parse_strings(src, "cudecompHaloCommBackendToString"),
        "config_defaults": cfg_defaults,
        "autotune_option_defaults": opt_defaults,
    }
    assert len(api["dtype_sizes"]) == 4 and len(api["transpose_backend_strings"]) == 8
    assert len(api["halo_backend_strings"]) == 5
    json.dump(api, open(os.path.join(OUT, "api_constants.json"), "w"), indent=1)
    print("wrote pencil_info.json shifted_rank.json api_constants.json")


if __name__ == "__main__":
    main()
