#!/usr/bin/env python3
"""Golden vector for gtx_vcf_header: the ##INFO / ##FORMAT / ##FILTER block Vcf::write_header prints
(/root/reference/src/typer/vcf.cpp:549-690), i.e. the reference's own output text, assembled from the string literals of that
function (adjacent literals concatenated as the compiler does).  Run in the build container only (reads /root/reference);
writes tests/golden/vcf_header_definitions.txt."""
import os
import re

SRC = "/root/reference/src/typer/vcf.cpp"
lines = open(SRC).read().split("\n")
begin = next(i for i, l in enumerate(lines) if "void Vcf::write_header" in l)
end = next(i for i in range(begin, len(lines)) if "// Column names" in lines[i])
text = []
for l in lines[begin:end]:
    s = l.strip()
    if s.startswith("//"):
        continue
    for m in re.finditer(r'"((?:[^"\\]|\\.)*)"', l):
        text.append(m.group(1).encode().decode("unicode_escape"))
blob = "".join(text)
defs = blob[blob.index("##INFO=<ID=AAScore"):]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vcf_header_definitions.txt")
open(out, "w").write(defs)
print(len(defs.split("\n")) - 1, "lines ->", out)
