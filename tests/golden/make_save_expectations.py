#!/usr/bin/env python3
"""Extracts the expectation tables of the reference's own container tests
(lib/test/TextureSaveTest.cpp:268-700: for DDS, KTX and PVR, which (format, type) pairs save
successfully and which are Unsupported) into tests/golden/save_expectations.json.  Runs only
where /root/reference is mounted; the JSON (data: names and outcomes) is what is committed."""
import json
import os
import re

SRC = "/root/reference/lib/test/TextureSaveTest.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "save_expectations.json")


def main():
    text = open(SRC).read()
    # expand the S3TC/ETC/ASTC/PVRTC helper macros in place
    macros = dict(re.findall(r"#define (\w+_SAVE_\w+_TESTS)\s*\\\n((?:.*\\\n)*.*)\n", text))
    table = {}
    for suite in re.finditer(r"INSTANTIATE_TEST_SUITE_P\(\w+,\s*TextureSave(Special)?(Dds|Ktx|Pvr)Test,(.*?)\)\);\n",
                             text, re.S):
        kind, body = suite.group(2).upper(), suite.group(3)
        for name, macro in macros.items():
            body = body.replace(name, macro)
        entries = table.setdefault(kind, {})
        for fmt, types in re.findall(r"TextureSaveTestInfo\(Texture::Format::(\w+),\s*\{(.*?)\}\)", body, re.S):
            for typ, outcome in re.findall(r"\{Texture::Type::(\w+),\s*(\w+)\}", types):
                entries["%s/%s" % (fmt, typ)] = outcome == "success"
    json.dump(table, open(OUT, "w"), indent=0, sort_keys=True)
    print({k: (len(v), sum(v.values())) for k, v in table.items()})


if __name__ == "__main__":
    main()
