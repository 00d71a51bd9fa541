"""Extracts the 256 values of the reference's PL binning table (include/graphtyper/typer/binned_pl.hpp, a constant array;
Vcf::write_record prints binned_pl[PL] and min(99, binned_pl[GQ])) into binned_pl.json.  Run in the build container:
    python tests/golden/make_binned_pl.py /root/reference
The JSON is data the tests compare the oracle's step function with (tests/test_vcf_text.py)."""
import json
import re
import sys

src = open(sys.argv[1] + "/include/graphtyper/typer/binned_pl.hpp").read()
body = src[src.index("binned_pl{") + len("binned_pl{"):src.rindex("}")]
vals = [int(x) for x in re.findall(r"^\s*(\d+)\s*,?\s*(?://.*)?$", body, flags=re.M)]
assert len(vals) == 256, len(vals)
json.dump({"source": "include/graphtyper/typer/binned_pl.hpp", "binned_pl": vals}, open(sys.argv[0].rsplit("/", 1)[0] + "/binned_pl.json", "w"))
print(len(vals), "values")
