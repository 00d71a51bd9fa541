"""TEST INFRASTRUCTURE: the structural-variant side of the reference's constructor restated line by line in Python, to
give the ORACLE its input records (VarRecord lists) for SV graphs.  The product has its own implementation
(graphtyper_amd/csrc/gtx_files.cpp), factored differently; tests compare the graph + index the two lead to.

Follows /root/reference/src/graph/constructor.cpp:
  transform_sv_records   :1079-1207     add_var_record (SV branch)  :1264-1491
  add_sv_breakend        :312-476       add_sv_deletion             :478-514
  add_sv_insertion       :515-725       add_sv_duplication          :727-871
  add_sv_inversion       :873-1031      append_sv_tag_to_node       :155-161
Parity unpinned: the reference's own SV constructor tests are commented out (test/graph/test_constructor.cpp:278-411)."""

E = 152  # EXTRA_SEQUENCE_LENGTH, constructor.cpp:1437
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


class Fasta:
    def __init__(self, seqs):
        self.seqs = {k: "".join(c if c in "ACGT" else "N" for c in v.upper()) for k, v in seqs.items()}  # seqan Dna5

    def read(self, chrom, begin, length):
        """read_reference_seq :245-257 (readRegion clips at the contig's ends)"""
        if length <= 0:
            return ""
        if begin < 0:
            length += begin
            begin = 0
        return self.seqs[chrom][begin:begin + max(length, 0)]

    def read_ends(self, chrom, begin, end, length):
        """read_reference_genome_ends :272-293"""
        if end - begin > 2 * length:
            return self.read(chrom, begin, length) + self.read(chrom, end - length, length)
        return self.read(chrom, begin, end - begin)


def complement(s):
    return "".join(COMP.get(c, c) for c in s)


def is_similar(a, b):
    """:1353-1394: global alignment score (match 1, mismatch -1, gap -1) of at most 1000 bases each / longer length >= 0.8"""
    if len(a) > 1000 and len(b) > 1000:
        a, b = a[:1000], b[:1000]
    larger = max(len(a), len(b))
    if larger == 0:
        return False
    prev = [-j for j in range(len(b) + 1)]
    for i in range(1, len(a) + 1):
        cur = [-i] + [0] * len(b)
        ai = a[i - 1]
        for j in range(1, len(b) + 1):
            cur[j] = max(prev[j - 1] + (1 if ai == b[j - 1] else -1), prev[j] - 1, cur[j - 1] - 1)
        prev = cur
    return prev[len(b)] / float(larger) >= 0.8


def transform_sv_records(fa, chrom, rec):
    """rec = dict(pos, ref, alt, info) of ONE alt; returns False when the record is dropped (:1079-1207)"""
    if len(rec["alt"]) == 0:
        return False
    if rec["pos"] == 0:
        return True
    if any(c in rec["alt"] for c in "<]["):
        return True
    size_diff = len(rec["alt"]) - len(rec["ref"])
    if size_diff <= -50:
        seq = ""
        if rec["ref"][0] != rec["alt"][0]:
            rec["pos"] -= 1
            rec["ref"] = fa.read(chrom, rec["pos"], 1)
            rec["alt"] = rec["ref"] + rec["alt"]
        if len(rec["alt"]) > 1:
            seq = rec["alt"][1:]
        add = ";" if len(rec["info"]) > 0 else ""
        add += "SVTYPE=DEL;SVLEN=%d;SVSIZE=%d;END=%d" % (-size_diff, -size_diff, len(seq) + rec["pos"] + 1 - size_diff)
        if seq:
            add += ";SEQ=" + seq
        rec["ref"] = rec["ref"][0]
        rec["alt"] = "<DEL>"
        rec["info"] += add
    elif size_diff >= 50:
        if rec["ref"][0] != rec["alt"][0]:
            rec["pos"] -= 1
            rec["ref"] = fa.read(chrom, rec["pos"], 1) + rec["ref"]
            seq = rec["alt"]
        else:
            seq = rec["alt"][1:]
        add = ";" if (len(rec["info"]) > 0 and rec["info"][-1] != ";") else ""
        add += "SVTYPE=INS;SVLEN=%d;SVSIZE=%d;SEQ=%s" % (size_diff, size_diff, seq)
        rec["alt"] = "<INS>"
        rec["info"] += add
    return True


class Builder:
    """state of one construct_graph call: the records made so far and Graph::SVs.size()"""

    def __init__(self, fa, chrom):
        self.fa, self.chrom, self.n_sv, self.records, self.svs = fa, chrom, 0, [], []

    def tag(self):
        return "<SV:%07d>" % self.n_sv

    def push_sv(self, sv, model, related=None):
        """graph.SVs.push_back(sv) with sv.model (and sv.related_sv: "next" / "prev") set just before, as the reference does it
        on its one SV object -- what is set stays set for the pushes behind it"""
        sv["model"] = model
        if related == "next":
            sv["related_sv"] = self.n_sv + 1
        elif related == "prev":
            sv["related_sv"] = self.n_sv - 1
        self.svs.append(dict(sv))
        self.n_sv += 1

    def rd(self, begin, length, chrom=None):
        return self.fa.read(chrom or self.chrom, begin, length)

    # ---- add_sv_breakend :312-476
    def breakend(self, var, alt, sv):
        var["ref"] = self.rd(var["pos"], 1)
        sv["original_alt"] = alt

        def chrom_name(c):
            return alt[alt.index(c) + 1:alt.rindex(":")]

        def position(c):
            colon = alt.rindex(":")
            return int(alt[colon + 1:alt.index(c, colon + 1)])

        if "[" in alt:
            at = alt.index("[")
            chrom2, pos = chrom_name("["), position("[")
            if at != 0:  # case 1
                bnd = var["ref"] + alt[1:at]
                bnd += self.rd(pos, E - len(bnd) + 1, chrom2)
                bnd += self.tag()
            else:  # case 2
                bnd = self.tag()
                second = alt.index("[", at + 1)
                seq = self.rd(pos - 1, E - (len(alt) - second), chrom2)
                bnd += complement(seq)[::-1] + alt[second + 1:]
        else:
            at = alt.index("]")
            chrom2, pos = chrom_name("]"), position("]")
            if at == 0:  # case 3
                bnd = self.tag()
                second = alt.index("]", at + 1)
                n = E - (len(alt) - second) - 1
                bnd += self.rd(pos - n, n, chrom2) + alt[second + 1:]
            else:  # case 4
                bnd = var["ref"] + alt[1:at]
                n = E - len(bnd) + 1
                bnd += complement(self.rd(pos - n, n, chrom2))[::-1]
                bnd += self.tag()
        var["alts"].append(bnd)
        self.push_sv(sv, sv["model"])

    # ---- add_sv_deletion :478-514
    def deletion(self, var, sv):
        var["ref"] = self.rd(var["pos"], 1)
        alt1 = var["ref"]
        if len(sv["seq"]) > 0 and sv["seq"][0] != ".":
            alt1 += sv["seq"]
        elif len(sv["ins_seq"]) > 0 and sv["ins_seq"][0] != ".":
            alt1 += sv["ins_seq"]
        if len(alt1) < E + 1:
            alt1 += self.rd(var["pos"] + len(sv["seq"]) + sv["size"] + 1, E + 1 - len(alt1))
        alt1 += self.tag()
        var["alts"].append(alt1)
        self.push_sv(sv, "BREAKPOINT")

    # ---- add_sv_insertion :515-725
    def insertion(self, var, sv, vcf_ref):
        var["ref"] = vcf_ref if vcf_ref[0] != "N" else self.rd(var["pos"], 1)
        if len(sv["seq"]) > 0:
            alt1 = self.rd(var["pos"], 1)
            alt2 = alt1
            if len(sv["seq"]) >= E:
                alt1 += sv["seq"][:E]
                alt1 += self.tag()
                self.push_sv(sv, "BREAKPOINT1", "next")
                alt2 += self.tag()
                alt2 += sv["seq"][-E:]
                self.push_sv(sv, "BREAKPOINT2", "prev")
            else:
                padding = E - len(sv["seq"])
                alt1 += sv["seq"]
                alt1 += self.rd(var["pos"] + 1, padding)
                alt1 += self.tag()
                self.push_sv(sv, "BREAKPOINT1", "next")
                alt2 += self.tag()
                alt2 += self.rd(var["pos"] - padding, padding + 1)
                alt2 += sv["seq"]
                self.push_sv(sv, "BREAKPOINT2", "prev")
            var["alts"] += [alt1, alt2]
        elif sv["or_start"] != -1 and sv["or_end"] != -1:
            alt1 = self.rd(var["pos"], 1)
            alt2 = ""
            ins = self.fa.read_ends(self.chrom, sv["or_start"] - 1, sv["or_end"], E)
            if len(ins) >= E:
                alt1 += ins[:E]
                alt1 += self.tag()
                self.push_sv(sv, "BREAKPOINT1", "next")
                alt2 += self.tag()
                alt2 += ins[-E:]
                self.push_sv(sv, "BREAKPOINT2", "prev")
            else:
                padding = E - len(ins)
                alt1 += ins
                alt1 += self.rd(var["pos"] + 1, padding)
                alt1 += self.tag()
                self.push_sv(sv, "BREAKPOINT1", "next")
                alt2 += self.tag()
                padding = min(padding, var["pos"])
                alt2 += self.rd(var["pos"] - padding, padding)
                alt2 += ins
                self.push_sv(sv, "BREAKPOINT2", "prev")
            var["alts"] += [alt1, alt2]
        elif len(sv["ins_seq_left"]) > 0 or len(sv["ins_seq_right"]) > 0:
            left, right = sv["ins_seq_left"][:E], sv["ins_seq_right"][:E]
            if len(left) > 1 and len(right) > 0:
                var["alts"].append(var["ref"] + left + self.tag())
                self.push_sv(sv, "BREAKPOINT1", "next")
                var["alts"].append(self.tag() + right)
                self.push_sv(sv, "BREAKPOINT2", "prev")
            elif len(left) > 1:
                var["alts"].append(var["ref"] + left + self.tag())
                self.push_sv(sv, "BREAKPOINT1")
            elif len(right) > 0:
                var["alts"].append(self.tag() + right)
                self.push_sv(sv, "BREAKPOINT2")

    # ---- add_sv_duplication :727-871
    def duplication(self, var, sv):
        var["ref"] = self.rd(var["pos"], 1)
        if sv["or_end"] == -1:
            if sv["or_start"] == -1:
                dup = self.fa.read_ends(self.chrom, var["pos"] + 1, var["pos"] + sv["length"] + 1, E)
                var2 = dict(var, alts=[])
                var["pos"] += sv["length"]
                var["ref"] = self.rd(var["pos"], 1)
                dup_begin = var["ref"] + sv["ins_seq"]
                dup_end = ""
                if len(dup) >= E:
                    dup_begin += dup[:E]
                    dup_begin += self.tag()
                    self.push_sv(sv, "BREAKPOINT1", "next")
                    dup_end += self.tag()
                    dup_end += dup[-E:]
                    dup_end += sv["ins_seq"]
                    self.push_sv(sv, "BREAKPOINT2", "prev")
                else:
                    padding = E - len(dup)
                    dup_begin += dup
                    dup_begin += self.rd(var["pos"] + 1, padding)
                    dup_begin += self.tag()
                    self.push_sv(sv, "BREAKPOINT1", "next")
                    padding = min(padding, var2["pos"])
                    dup_end += self.tag()
                    dup_end += self.rd(var2["pos"] - padding + 1, padding)
                    dup_end += dup
                    self.push_sv(sv, "BREAKPOINT2", "prev")
                var["alts"].append(dup_begin)
                var2["alts"].append(dup_end)
                self.records.append(var2)
            else:
                dup_begin = var["ref"] + sv["ins_seq"]
                dup_begin += self.rd(sv["or_start"] - 1, E)
                dup_begin += self.tag()
                var["alts"].append(dup_begin)
                self.push_sv(sv, "BREAKPOINT1")
        else:
            start = max(E, sv["or_end"])
            dup_begin = self.tag()
            dup_begin += self.rd(start - E, E)
            dup_begin += sv["ins_seq"]
            var["alts"].append(dup_begin)
            self.push_sv(sv, "BREAKPOINT2")

    # ---- add_sv_inversion :873-1031
    def inversion(self, var, sv):
        var["ref"] = self.rd(var["pos"], 1)
        if sv["inv_type"] == "INV3":
            sv["or_end"] = sv["end"]
        elif sv["inv_type"] == "INV5":
            sv["or_start"] = sv["begin"]
            sv["begin"] += sv["size"]
            var["pos"] += sv["size"]
            var["ref"] = self.rd(var["pos"], 1)
        if sv["or_end"] == -1:
            if sv["or_start"] == -1:
                dup = self.fa.read_ends(self.chrom, var["pos"] + 1, var["pos"] + sv["length"] + 1, E)
                inv = complement(dup[::-1])
                inv_begin = var["ref"] + sv["ins_seq"]
                var2 = dict(var, alts=[])
                var2["pos"] += sv["length"]
                var2["ref"] = self.rd(var2["pos"], 1)
                inv_end = ""
                if len(inv) >= E:
                    inv_begin += inv[:E]
                    inv_begin += self.tag()
                    self.push_sv(sv, "BREAKPOINT1", "next")
                    inv_end += self.tag()
                    inv_end += inv[-E:]
                    inv_end += sv["ins_seq"]
                    self.push_sv(sv, "BREAKPOINT2", "prev")
                else:
                    padding = E - len(inv)
                    inv_begin += inv
                    inv_begin += self.rd(var["pos"] + 1, padding)
                    inv_begin += self.tag()
                    self.push_sv(sv, "BREAKPOINT1", "next")
                    padding = min(padding, var2["pos"])
                    inv_end += self.tag()
                    inv_end += self.rd(var2["pos"] - padding + 1, padding)
                    inv_end += inv
                    inv_end += sv["ins_seq"]
                    self.push_sv(sv, "BREAKPOINT2", "prev")
                var["alts"].append(inv_begin)
                var2["alts"].append(inv_end)
                self.records.append(var2)
            else:
                dup = complement(self.rd(sv["or_start"] - 1, E))
                inv = self.tag() + dup[::-1] + sv["ins_seq"]
                var["alts"].append(inv)
                self.push_sv(sv, "BREAKPOINT2")
        else:
            start = max(E, sv["or_end"])
            dup = complement(self.rd(start - E, E))
            inv = var["ref"] + sv["ins_seq"] + dup[::-1]
            inv += self.tag()
            var["alts"].append(inv)
            self.push_sv(sv, "BREAKPOINT1")

    # ---- the SV branch of add_var_record :1264-1491
    def add_sv(self, pos, vcf_ref, alt, info, vcf_id=""):
        var = dict(pos=pos, ref="", alts=[], is_sv=True)
        sv = dict(type=None, chrom=self.chrom, begin=pos + 1, end=0, size=0, length=0, n_clusters=0, num_merged_svs=-1, or_start=-1, or_end=-1,
                  related_sv=-1, model="AGGREGATED", old_variant_id=vcf_id, seq="", hom_seq="", ins_seq="", ins_seq_left="", ins_seq_right="",
                  inv_type=None, original_alt="")
        is_a_dup = False
        for item in info.split(";"):
            key, _, val = item.partition("=")
            if key == "DUPSVLEN":
                is_a_dup = True
            if key == "SVTYPE":
                sv["type"] = {"DEL": "DEL", "DEL:ME:ALU": "DEL_ALU", "DUP": "DUP", "INV": "INV", "INS": "INS", "INS:ME:ALU": "INS_ALU",
                              "BND": "BND"}.get(val, "OTHER")
            elif key in ("END", "SVSIZE", "SVLEN", "ORSTART", "OREND", "NCLUSTERS", "NUM_MERGED_SVS"):
                sv[{"END": "end", "SVSIZE": "size", "SVLEN": "length", "ORSTART": "or_start", "OREND": "or_end", "NCLUSTERS": "n_clusters",
                    "NUM_MERGED_SVS": "num_merged_svs"}[key]] = int(val)
            elif key in ("SEQ", "SVINSSEQ", "LEFT_SVINSSEQ", "RIGHT_SVINSSEQ", "DUPSVINSSEQ"):
                if len(val) > 0 and val[0] != ".":
                    sv[{"SEQ": "seq", "SVINSSEQ": "ins_seq", "LEFT_SVINSSEQ": "ins_seq_left", "RIGHT_SVINSSEQ": "ins_seq_right",
                        "DUPSVINSSEQ": "ins_seq"}[key]] = val
            elif key in ("INV3", "INV5"):
                sv["inv_type"] = key
        assert sv["type"] is not None, "SV without SVTYPE"
        if sv["type"] == "INS" and is_a_dup:
            sv["type"] = "DUP"
        if sv["length"] < 0:
            sv["length"] = -sv["length"]
        if sv["type"] != "BND" and sv["length"] == 0:
            sv["length"] = sv["size"]
            if sv["length"] == 0:
                sv["length"] = len(sv["seq"])
                if sv["length"] == 0:
                    sv["length"] = len(sv["ins_seq"])
        if sv["size"] == 0:
            sv["size"] = sv["length"]
        if sv["end"] == 0:
            sv["end"] = sv["begin"] + sv["size"]
        if sv["type"] == "INS" and len(sv["seq"]) > 0:
            n = len(sv["seq"])
            if var["pos"] - 1 - n >= 0:
                if is_similar(self.rd(max(0, var["pos"] - 1 - n), n), sv["seq"]):
                    var["pos"] -= n
                    sv["type"] = "DUP"
            if sv["type"] == "INS":
                if is_similar(self.rd(var["pos"] + 1, n), sv["seq"]):
                    sv["type"] = "DUP"
        if sv["type"] == "BND":
            self.breakend(var, alt, sv)
        elif sv["type"] in ("DEL", "DEL_ALU"):
            self.deletion(var, sv)
        elif sv["type"] == "DUP":
            self.duplication(var, sv)
        elif sv["type"] == "INS":
            self.insertion(var, sv, vcf_ref)
        elif sv["type"] == "INV":
            self.inversion(var, sv)
        else:
            return
        if len(var["alts"]) > 0:
            self.records.append(var)


def sv_table_text(svs):
    """Graph::SVs as the text the oracle (gto_sv.hpp: parse_sv_table) and gtx_graph_sv_table share: one SV per line, tab
    separated, "." for an empty field"""
    def f(x):
        x = "" if x is None else str(x)
        return x if x != "" else "."
    cols = ("type", "chrom", "begin", "length", "size", "end", "n_clusters", "num_merged_svs", "or_start", "or_end", "related_sv", "model",
            "old_variant_id", "inv_type", "seq", "hom_seq", "ins_seq", "ins_seq_left", "ins_seq_right", "original_alt")
    return "".join("\t".join(f(sv[c]) for c in cols) + "\n" for sv in svs)


def sv_records(seqs, vcf_lines, chrom, region_begin=0, region_end=0xFFFFFFFF, with_table=False):
    """construct_graph's record intake for an SV graph (:1650-1760): -> [(pos0, ref, [alts], info)] sorted by position, as the
    oracle takes them ("SV=1" marks VarRecord::is_sv).  vcf_lines: tab-separated VCF data lines."""
    fa = Fasta(seqs)
    b = Builder(fa, chrom)
    for line in vcf_lines:
        if not line.strip() or line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        if f[0] != chrom:
            continue
        pos0, ref, info = int(f[1]) - 1, f[3], (f[7] if len(f) > 7 else "")
        if pos0 < region_begin or pos0 + len(ref) > region_end or not ref or not f[4]:
            continue
        for alt in f[4].split(","):
            if not alt or alt[0] == ".":
                continue
            rec = dict(pos=pos0, ref=ref, alt=alt, info=info)
            if not transform_sv_records(fa, chrom, rec):
                continue
            if len(rec["alt"]) >= 5 and any(c in rec["alt"] for c in "<[]"):
                b.add_sv(rec["pos"], rec["ref"], rec["alt"], rec["info"], f[2] if len(f) > 2 else "")
            elif all(c in "ACGT" for c in rec["alt"]):
                b.records.append(dict(pos=rec["pos"], ref=rec["ref"], alts=[rec["alt"]], is_sv=False))
    out = sorted(b.records, key=lambda r: r["pos"])  # (stable; the reference's std::sort need not be for equal positions)
    recs = [(r["pos"], r["ref"], r["alts"], "SV=1" if r["is_sv"] else ".") for r in out]
    return (recs, sv_table_text(b.svs)) if with_table else recs
