#!/usr/bin/env python3
"""Extracts the (format, type) pairs the reference's own TextureConvertTest /
TextureConvertSpecialTest instantiate (lib/test/TextureTest.cpp:869-985: every pair is expected
to convert) into tests/golden/convert_expectations.json.  Runs only where /root/reference is
mounted; the JSON (names) is what is committed."""
import json
import os
import re

SRC = "/root/reference/lib/test/TextureTest.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "convert_expectations.json")


def main():
    text = open(SRC).read()
    macros = dict(re.findall(r"#define (\w+_CONVERSION_TESTS)\s*\\\n((?:.*\\\n)*.*)\n", text))
    pairs = set()
    for suite in re.finditer(r"INSTANTIATE_TEST_SUITE_P\(\w+,\s*TextureConvert(?:Special)?Test,(.*?)\)\);\n", text, re.S):
        body = suite.group(1)
        for name, macro in macros.items():
            body = body.replace(name, macro)
        for fmt, types in re.findall(r"TextureConvertTestInfo\(Texture::Format::(\w+),\s*\{(.*?)\}\)", body, re.S):
            for typ in re.findall(r"Texture::Type::(\w+)", types):
                pairs.add("%s/%s" % (fmt, typ))
    json.dump(sorted(pairs), open(OUT, "w"), indent=0)
    print(len(pairs))


if __name__ == "__main__":
    main()
