#!/usr/bin/env python3
"""Golden g15: the command-line surface of the reference's two training scripts -- (flag, type, default-kind) triples scanned from
/root/reference/tools/trainV2_simt.py:72-157 and trainV1_warmup.py:60-150 (data only: names and kinds, no source text).
tests/test_host_logic.py checks that simt_amd.tools.{trainV2_simt,trainV1_warmup}.get_arguments accept every one of them."""
import json
import os
import re

REF = "/root/reference/tools"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g15_cli_flags.json")


def scan(path):
    src = open(path).read()
    flags = []
    for m in re.finditer(r'add_argument\(\s*"(--[\w-]+)"\s*,\s*([^)]*?)\)', src, re.S):
        body = m.group(2)
        kind = "store_true" if "store_true" in body else (re.search(r"type=(\w+)", body) or [None, "str"])[1]
        flags.append([m.group(1), kind])
    return flags


if __name__ == "__main__":
    out = {"trainV2_simt": scan(os.path.join(REF, "trainV2_simt.py")), "trainV1_warmup": scan(os.path.join(REF, "trainV1_warmup.py"))}
    json.dump(out, open(OUT, "w"), indent=0)
    print({k: len(v) for k, v in out.items()})
