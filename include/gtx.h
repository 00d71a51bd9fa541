/* gtx.h -- C ABI of libgtx.so: MI355X-native read -> pangenome-graph alignment + genotype scoring.
 *
 * The reference (graphtyper v2.7.7) has no FFI seam; this header is the boundary a maintainer would call
 * from the three C++ sites the hot path sits behind:
 *
 *   gtx_ctx_create      replaces  PHIndex index_graph(Graph const&)            include/graphtyper/index/indexer.hpp:16
 *                       + the upload of the immutable global `gyper::graph`     include/graphtyper/graph/graph.hpp:171
 *   gtx_align_batch     replaces  align_read(bam1_t*, seq, rseq, PHIndex const&) include/graphtyper/typer/alignment.hpp:15-18
 *   gtx_score_batch     replaces  update_unpaired_read_paths / update_paths / get_better_paths
 *                                                                               include/graphtyper/typer/alignment.hpp:20-33
 *                       and       VcfWriter::update_haplotype_scores_geno (both overloads)
 *                                                                               include/graphtyper/typer/vcf_writer.hpp:33-41
 *   gtx_scores_finalize replaces  reading VcfWriter::haplotypes[*].hap_samples[*] / var_stats after the read loop
 *                                                                               src/utilities/hts_parallel_reader.cpp:782-1033
 *   gtx_stream_*        mirrors   the per-record logic of parallel_reader_genotype_only / genotype_only
 *                                 (flag filter, duplicate reuse, mate parking; SV calling: record filter, coverage
 *                                 filter, leftover reads)                       src/utilities/hts_parallel_reader.cpp:245-338,528-772
 *   gtx_phase_flags     replaces  the `ph` construction                         src/utilities/hts_parallel_reader.cpp:782-904
 *   gtx_scores_replay   replays   the saturation guard of Haplotype::explain_to_score     src/graph/haplotype.cpp:560
 *   gtx_scores_reduce   replaces  the merge of the per-thread / per-pool results   src/typer/caller.cpp:439-482,
 *                                 (every per-read effect is an integer addition)  src/typer/vcf_operations.cpp:366-374
 *   gtx_vcf_records     replaces  Vcf::add_haplotype + generate_infos + write_record   src/typer/vcf.cpp:767-1151,1507-1611
 *   gtx_reads_*         replaces  HtsReader / HtsParallelReader for BAM files            src/utilities/hts_reader.cpp:17-303
 *   gtx_bam_shrink      replaces  bamshrink (the read pre-filter)                        src/utilities/bamshrink.cpp:667-1371
 *   gtx_graph_build     replaces  Graph::add_genomic_region                     src/graph/graph.cpp:41-339
 *   gtx_graph_from_files replaces construct_graph (small variants, structural variants) src/graph/constructor.cpp:1597-1777
 *
 * Conventions: plain pointers and sizes only; the caller allocates and owns every buffer; a context is immutable
 * after creation and may be used from several host threads; every function returns a status code (0 = ok) and never
 * aborts; results for a read do not depend on batch composition or order.  All compute runs on the GPU -- there is no
 * CPU fallback: without a usable HIP device gtx_ctx_create fails with GTX_ERR_NO_DEVICE.
 */
#ifndef GTX_H
#define GTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTX_INVALID_ID 0xFFFFFFFFu
#define GTX_SPECIAL_START 0xD0000000u /* include/graphtyper/constants.hpp.in:33 */
#define GTX_K 32

enum
{
  GTX_OK = 0,
  GTX_ERR_ARG = 1,         /* NULL / inconsistent arguments */
  GTX_ERR_NO_DEVICE = 2,   /* no HIP device: the product has no CPU path */
  GTX_ERR_HIP = 3,         /* a HIP runtime call failed (see gtx_last_error) */
  GTX_ERR_UNSUPPORTED = 4, /* outside the supported envelope (a site with more alleles than MAX_NUMBER_OF_HAPLOTYPES, VCF text of an SV graph, ...) */
  GTX_ERR_CAPACITY = 5,    /* a caller-provided buffer is too small */
  GTX_ERR_GRAPH = 6,       /* malformed graph view */
  GTX_ERR_IO = 7           /* a file could not be opened or is truncated / malformed */
};

/* per read x orientation status bits written by gtx_align_batch (word 0, bits 16..31 of a result record) */
enum
{
  GTX_ST_LABEL_OVERFLOW = 1, /* one k-mer list returned more labels than the kernel's staging buffer */
  GTX_ST_PATH_OVERFLOW = 2,  /* more live paths / variant sites per path than the kernel's tables */
  GTX_ST_DFS_OVERFLOW = 4,   /* graph walk produced more candidate sequences than the kernel's table */
  GTX_ST_RECORD_OVERFLOW = 8, /* result fits neither rec_words nor the context's big-record arena */
  GTX_ST_EXTERNAL = 16        /* not an error: the path words of this record are in the big-record arena (see gtx_align_batch) */
};
#define GTX_ST_ERROR_MASK 15u
/* allele sets of a record: two words (alleles 0..63), or -- in a record whose second word carries GTX_REC_WIDE -- this many
 * words (MAX_NUMBER_OF_HAPLOTYPES = 2560 alleles, include/graphtyper/constants.hpp.in:23) */
#define GTX_WIDE_MASK_WORDS 80u
#define GTX_REC_HAS_VARIANTS 0x80000000u /* bit 31 of record word 1: some path of the record carries a variant site */
#define GTX_REC_WIDE 0x40000000u         /* bit 30 of record word 1: every site of the record is (hap, GTX_WIDE_MASK_WORDS mask words)
                                            instead of (hap, mask_lo, mask_hi) -- a path names an allele >= 64 */

/* Graph as SoA node tables = the reference's Graph::ref_nodes / var_nodes (include/graphtyper/graph/graph.hpp:40-134):
 * strictly alternating  ref node r -> its ref_nvar[r] var nodes (allele 0 = reference allele) -> ref node r+1.
 * Orders are 1-based contig positions (Label::order).  dna holds every node's bases back to back.
 * events: per var node two sets (events, anti_events) in CSR form; may be NULL when no node has events. */
typedef struct gtx_graph_view
{
  uint32_t n_ref, n_var;
  const uint32_t * ref_order;     /* [n_ref] Label::order */
  const uint32_t * ref_len;       /* [n_ref] Label::dna.size() */
  const uint32_t * ref_dna_off;   /* [n_ref] offset into dna */
  const uint32_t * ref_nvar;      /* [n_ref] RefNode::out_degree() */
  const uint32_t * ref_first_var; /* [n_ref] RefNode::get_var_index(0) (ignored when ref_nvar == 0) */
  const uint32_t * var_order;     /* [n_var] */
  const uint32_t * var_len;       /* [n_var] */
  const uint32_t * var_dna_off;   /* [n_var] */
  const uint32_t * var_out_ref;   /* [n_var] VarNode::out_ref_id */
  const char * dna;
  uint64_t dna_len;
  const uint32_t * event_off; /* [2*n_var+1] or NULL: events of var v = event_val[event_off[2v] .. event_off[2v+1]),
                                 anti events = event_val[event_off[2v+1] .. event_off[2v+2]) */
  const int64_t * event_val;
} gtx_graph_view;

/* ---- host graph builder: variant records + reference sequence -> node tables ----
 * replaces Graph::add_genomic_region(std::vector<char>&&, std::vector<VarRecord>&&, GenomicRegion&&)
 *          include/graphtyper/graph/graph.hpp:60-62, src/graph/graph.cpp:41-339 (filters, record merging, node emission)
 * and, with extend_prefix, GenomicRegion::add_reference_to_record_if_they_have_a_matching_prefix
 *          src/graph/genomic_region.cpp:236-256 (what the constructor applies to every VCF record, constructor.cpp:1740-1744) */
typedef struct gtx_allele
{
  const char * seq;
  uint32_t len;
  const int64_t * events; /* Ref/Alt::events (include/graphtyper/graph/alt.hpp:19-20) */
  uint32_t n_events;
  const int64_t * anti_events;
  uint32_t n_anti_events;
} gtx_allele;

typedef struct gtx_record /* VarRecord (include/graphtyper/graph/var_record.hpp:15-20); alleles[0] is REF */
{
  uint32_t pos; /* 0-based contig position */
  uint32_t n_alleles;
  const gtx_allele * alleles;
  int32_t is_sv;
} gtx_record;

typedef struct gtx_graph gtx_graph; /* owns the node tables a gtx_graph_view points into */

/* records sorted by pos; region = [region_begin, region_end) 0-based, reference[0] is contig position region_begin.
 * GTX_ERR_ARG when the records are not sorted or one that starts inside the region runs past the reference sequence. */
int gtx_graph_build(const char * reference, uint64_t reference_len, int64_t region_begin, int64_t region_end,
                    const gtx_record * records, uint32_t n_records, int add_all_variants, int is_sv_graph, int extend_prefix,
                    gtx_graph ** out);
/* Graph of one region straight from files: replaces construct_graph(reference_filename, vcf_filename, region, is_sv_graph,
 * use_index) (include/graphtyper/graph/constructor.hpp, src/graph/constructor.cpp:1597-1777) with split_multi_allelic
 * (:1033-1077), the small-variant branch of add_var_record (:1208-1262, 1493-1595) and, in an SV graph, the structural
 * variant branches: breakends (:312-476), <DEL> (:478-514), <INS> (:515-725), <DUP> (:727-871), <INV> (:873-1031) and
 * transform_sv_records (:1079-1206).  fasta_path: plain FASTA, its .fai is used when present;
 * vcf_path: plain or gzip/bgzip VCF, NULL or "" for a reference-only graph; region: "chr", "chr:begin" or "chr:begin-end"
 * (1-based, GenomicRegion, src/graph/genomic_region.cpp:73-113).  region_begin / region_end (may be NULL) receive the
 * 0-based span [begin, end) of the reference bases that were read. */
int gtx_graph_from_files(const char * fasta_path, const char * vcf_path, const char * region, int add_all_variants, int is_sv_graph,
                         gtx_graph ** out, int64_t * region_begin, int64_t * region_end);
int gtx_graph_get_view(const gtx_graph *, gtx_graph_view * out);
/* Graph::SVs (include/graphtyper/graph/sv.hpp:36-63) of a graph made by gtx_graph_from_files with structural variants, as
 * text: one SV per line in the order of the <SV:nnnnnnn> tags of the allele sequences, tab separated: type chrom begin length
 * size end n_clusters num_merged_svs or_start or_end related_sv model old_variant_id inv_type seq hom_seq ins_seq
 * ins_seq_left ins_seq_right original_alt ("." for an empty field).  gtx_vcf_records needs it for the calls of an SV graph.
 * Writes min(*len, cap) bytes (out may be NULL with cap 0 to ask for the length). */
int gtx_graph_sv_table(const gtx_graph *, char * out, uint64_t cap, uint64_t * len);
void gtx_graph_destroy(gtx_graph *);

/* Options the path reads (include/graphtyper/utilities/options.hpp:34,82,87,89,90) */
typedef struct gtx_params
{
  int32_t max_index_labels;              /* 75 */
  int32_t is_sv_graph;                   /* Graph::is_sv_graph */
  int32_t hq_reads;                      /* Options::hq_reads */
  int32_t force_align_both_orientations; /* Options::force_align_both_orientations */
  int32_t is_segment_calling;            /* Options::is_segment_calling */
  int32_t sam_flag_filter;               /* 3840 */
  int32_t no_second_pass;                /* 1: reads that overflow the main pass' tables keep their status bit (A/B tests) */
  uint32_t big_record_words;             /* capacity of the big-record arena in uint32 words, 0 = 16 Mi */
  uint32_t exact_pass_mb;                /* MiB of HBM per slab of the exact alignment pass, of which a context makes one per batch in flight, up to four (gtx_align_batch: the
                                          * pass whose tables have no fixed size), 0 = the GTX_EXACT_PASS_MB environment
                                          * variable, else 2048 (4096 for a graph with a site of more than 64 alleles): up to 1 024 tasks of a repeat side by side */
} gtx_params;

/* One KmerLabel (include/graphtyper/index/kmer_label.hpp:13-41) */
typedef struct gtx_label
{
  uint32_t start_index, end_index, variant_id;
} gtx_label;

/* Longest read the kernels align (the reference has no limit; its MAX_READ_LENGTH constant, constants.hpp.in:27, is 151).
 * A longer read gets empty records carrying GTX_ST_RECORD_OVERFLOW in every pass, and gtx_stream_push refuses a batch
 * that holds one (GTX_ERR_UNSUPPORTED) instead of letting it vanish from the accumulators. */
#define GTX_MAX_READ 256

/* Per read fields of bam1_t the path looks at (src/typer/alignment.cpp:331-363) */
typedef struct gtx_read_meta
{
  uint16_t l_qseq; /* bases */
  uint16_t flag;
  int32_t tid, mtid;
  int32_t isize;
  /* bam core.pos of the record (0-based contig position of the first aligned base) minus its leading soft clip, i.e.
   * where read base 0 would lie on the reference if the mapper was right; -1 = unknown.  A HINT ONLY: the kernels use
   * it to look at the index entries of that place first (reads of a sorted BAM share them) and prove from flags stored
   * there that the global lookups of the reference (ph_index.cpp:66-107) would return the same; whatever cannot be
   * proven is looked up globally.  Results never depend on it (tests feed wrong and missing hints). */
  int32_t pos;
} gtx_read_meta;

/* gtx_rec_meta::flag, besides the SAM bits: the reverse orientation of the read this record uses was not aligned
 * (align_read aligns forward only for unpaired reads and concordant pairs, alignment.cpp:341-352), its record is empty
 * by construction and the scorer does not fetch it.  Set by gtx_stream_push; optional for hand-made items.
 * The same bit in gtx_read_meta::flag (gtx_stream_push sets it there as well) is the caller's promise never to look at the
 * reverse record of that read when align_read does not ask for the reverse orientation: the alignment then does not write
 * that record's empty header (its slot keeps what it held), and every item that uses the task has to carry the bit too. */
#define GTX_FLAG_FORWARD_ONLY 0x8000u

/* Per record fields consumed by update_unpaired_read_paths / update_paths (src/typer/alignment.cpp:365-545) */
typedef struct gtx_rec_meta
{
  uint32_t align_index; /* which gtx_align_batch result this record uses (duplicate reads reuse their predecessor's) */
  uint16_t flag;
  uint8_t mapq;
  uint8_t score_diff; /* AS-XS as get_score_diff() computes it (src/typer/alignment.cpp:140-325) */
  int32_t pos;        /* bam core.pos */
  int32_t isize;
} gtx_rec_meta;

/* One call of genotype_only() that reaches the VcfWriter: an unpaired record (second.align_index == GTX_INVALID_ID)
 * or a mate pair (first = the parked mate, second = the record that completed the pair).
 * kind GTX_ITEM_LEFTOVER (SV calling, src/utilities/hts_parallel_reader.cpp:719-745): a read whose mate never came;
 * second = the same record with IS_FIRST_IN_PAIR | IS_SEQ_REVERSED toggled, the better orientation pair is chosen as for
 * mates but only its first member is scored, alone. */
#define GTX_ITEM_LEFTOVER 1u
typedef struct gtx_score_item
{
  gtx_rec_meta first, second;
  uint32_t sample; /* pn_index */
  uint32_t kind;   /* 0 or GTX_ITEM_LEFTOVER */
} gtx_score_item;

typedef struct gtx_ctx gtx_ctx;

/* sizes of the score accumulators for this graph */
typedef struct gtx_score_layout
{
  uint32_t n_hap;        /* = number of variant sites = Graph::get_all_haplotypes().size() */
  uint64_t total_tri;    /* sum over haplotypes of cnum*(cnum+1)/2 */
  uint64_t total_allele; /* sum over haplotypes of cnum */
  uint64_t total_near;   /* connection counters between near haplotypes (gtx_ctx_near_pairs) */
  uint32_t ref_depth_len; /* positions of the region's reference (Graph::reference.size()): length of the depth track */
} gtx_score_layout;

const char * gtx_strerror(int status);
const char * gtx_last_error(void); /* thread local detail of the last failing call */

/* Builds the k-mer index of the graph (index_graph), flattens graph + index and uploads both to `device`. */
int gtx_ctx_create(const gtx_graph_view * graph, const gtx_params * params, int device, gtx_ctx ** out);
void gtx_ctx_destroy(gtx_ctx *);

/* Device memory of contexts, their scratch and gtx_scores_alloc blocks comes from a process-wide cache: what a destroyed
 * context held is handed to the next one instead of going back to the driver (a region's context lives for milliseconds, and
 * hipMalloc / hipFree were most of what creating one cost).  The cache keeps at most GTX_DEVICE_CACHE_MB (environment, default
 * 32768) and is emptied by this call. */
void gtx_device_cache_release(void);

/* Graph facts derived at creation (Graph::create_special_positions, graph.cpp:384-407). out arrays may be NULL. */
int gtx_ctx_special_positions(const gtx_ctx *, uint32_t * n_special, uint32_t * ref_reach_poses, uint32_t * actual_poses,
                              uint32_t cap);
int gtx_ctx_score_layout(const gtx_ctx *, gtx_score_layout * out);
/* hap_order[n_hap], hap_cnum[n_hap], tri_off[n_hap], allele_off[n_hap] */
int gtx_ctx_haplotypes(const gtx_ctx *, uint32_t * hap_order, uint32_t * hap_cnum, uint64_t * tri_off, uint64_t * allele_off);
/* layout of d_conn_near: near_last[n_hap], near_off[n_hap] */
int gtx_ctx_near_pairs(const gtx_ctx *, uint32_t * near_last, uint64_t * near_off);

/* Index inspection (host copy, reference order): PHIndex::get(key) */
int gtx_index_stats(const gtx_ctx *, uint64_t * n_keys, uint64_t * n_labels);
int gtx_index_get(const gtx_ctx *, uint64_t key, gtx_label * out, uint32_t cap, uint32_t * n);
/* keys ascending, counts per key, labels in bucket order */
int gtx_index_dump(const gtx_ctx *, uint64_t * keys, uint32_t * counts, gtx_label * labels);
/* Inspection of the position-hinted pass' tables as the context holds them (on its device, or on the host for an
 * inspection-only context): which = 0 per-position flags (2 x u32 per position), 1 the linear reference as bit planes,
 * 2 the site behind every position's node (2 x u32), 3 / 4 the filters over the first / last halves of the indexed keys,
 * 5 the allele windows (8 x u32 each: site, allele, bases of the allele, bases of the site's reference allele, order of the
 * site's variant nodes, 3 unused) whose positions continue tables 0..2 behind the linear reference, 6 per reference node the
 * first window of the site behind it | number of its windows << 24 (both empty for a graph without windows).
 * out == NULL: only *bytes.  Tests compare the device's build with the host's (gtx_index_dev.hip / gtx_host.cpp: the
 * reference has no such tables -- they restate what its index lookups would return at the place the mapper named). */
int gtx_ctx_hint_table(const gtx_ctx *, int which, void * out, uint64_t cap_bytes, uint64_t * bytes);

/* ---- device entry points.  All `d_` pointers are DEVICE pointers owned by the caller. ----
 *
 * d_seq      : n_reads * seq_stride bytes, BAM 4-bit packed bases (bam_get_seq layout: high nibble first)
 * d_meta     : n_reads gtx_read_meta
 * d_records  : n_reads * 2 * rec_words uint32; record (read i, orientation o) starts at (2*i+o)*rec_words:
 *    w0 = n_paths | status << 16, w1 = longest_path_length | l_qseq << 16 | GTX_REC_HAS_VARIANTS, then per path
 *    start, end, read_start_index | read_end_index << 16, mismatches | n_var << 16, n_var * (hap, mask_lo, mask_hi)
 *    hap = haplotype (variant site) index, Path::var_order = hap_order[hap] of gtx_ctx_haplotypes;
 *    mask bit a set <=> allele a in Path::nums; in a record with GTX_REC_WIDE in w1 a site is (hap, GTX_WIDE_MASK_WORDS
 *    mask words) -- only graphs with a site of more than 64 alleles produce those, and they are always GTX_ST_EXTERNAL
 *    A record with GTX_ST_EXTERNAL (a result with more paths than rec_words holds) keeps w0/w1 and has w2 = word offset
 *    of its path words in the context's big-record arena (gtx_ctx_big_records).  Like d_records the arena is meant to
 *    stay resident while a region is processed (a parked mate is scored batches later): it only grows until
 *    gtx_ctx_big_records_rewind; when it is full such a read ends with GTX_ST_RECORD_OVERFLOW.
 * stream     : hipStream_t or NULL
 * Passes: the main kernel keeps a read's tables in LDS; the few reads that exceed them (repeats: hundreds of seed
 * locations) are queued on the device and redone by a second kernel over larger tables in HBM (512 paths, 2048 labels
 * per k-mer), and what exceeds those as well -- the reference has NO limit on the paths and labels of a read
 * (src/typer/genotype_paths.cpp:294-352: a read inside a 280-bp homopolymer chains 249 x 249 labels) -- by the exact pass,
 * whose tables are cut at run time out of a slab of HBM (gtx_params::exact_pass_mb; a context makes one per batch in flight, up to four; first a small
 * part of the slab per task, then a large one, then all of it).  Every read therefore gets the result the reference computes; an overflow status (and no
 * paths) is left only on a read that needs more than the configured slab, or whose record finds the big-record arena full
 * (GTX_ST_RECORD_OVERFLOW; a record counts its paths in 16 bits).  gtx_ctx_exact_pass_tasks tells how many tasks went that far.
 * In front of them runs the position-hinted pass (gtx_read_meta::pos): one read per lane, finished there when the flags of
 * the hinted place prove what the global lookups would return; everything else goes on to the passes above unchanged.
 * (Two builds of it: graphs whose sites lie within a k-mer of each other get the one that also takes k-mers over two sites,
 * walks at the read's end over sites with alleles of any length, and reads that carry another allele than the reference's --
 * judged on that allele's path, whose tables continue the linear reference's: gtx_ctx_hint_table 5 / 6.  Rows of more than
 * 80 bytes -- reads of up to 256 bases -- get a third build with eight k-mers.)
 * Re-entrant: calls on one context may overlap in time from several host threads and streams, like align_read is called
 * from the reference's worker threads (src/typer/caller.cpp:399-436); each call draws its queues, counters and
 * workspaces from a pool inside the context. */
int gtx_align_batch(gtx_ctx *, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                    uint32_t * d_records, uint32_t rec_words, void * stream);

/* gtx_align_batch with a dense side array: d_task_flags[2 * read + orientation] (one byte per task, indexed like d_records,
 * so the caller offsets the pointer per batch the same way) receives GTX_TASK_HAS_VARIANTS when the task's record carries a
 * variant site -- the one fact the first stage of gtx_score_batch needs of nearly every record.  With it that stage reads
 * one byte per read instead of one cache line per record header (gtx_score_batch_flags).  NULL: plain gtx_align_batch. */
#define GTX_TASK_HAS_VARIANTS 1u
int gtx_align_batch_flags(gtx_ctx *, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                          uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream);

/* ---- reads as bit planes: the layout the alignment kernels read.  north_star asks for "coalesced HBM loads of packed
 * 2-bit reads"; BAM holds 4-bit codes (bam_get_seq: A=1 C=2 G=4 T=8, N=15 and the other IUPAC sets, '='=0), and the path must
 * see every one of them (to_uint64_vec fans ambiguity codes out, src/utilities/type_conversions.cpp:207-266).  A plane row
 * keeps all four bits, one plane each: a row of plane_stride bytes (a multiple of 16) is plane_stride / 16 groups of 32 bases,
 * four little-endian 32-bit words per group -- word 4g + b holds bit b of the codes of bases 32g .. 32g+31, base 32g + j at
 * bit j; bits behind the read's end are ignored.  For an unambiguous read planes 1|3 and 2|3 ARE the packed 2-bit bases (low
 * and high bit of A0 C1 G2 T3), plane 0&1&2&3 is the N mask.  Same size as the BAM nibbles (80 bytes for 150 bp), but the per-base
 * questions of the kernels become bitwise operations over 32 bases and nothing is transposed per read and step.
 *   gtx_pack_planes        host: n BAM nibble rows (seq_stride bytes each) -> n plane rows
 *   gtx_stream_set_planes  gtx_stream_push then writes align_seq as plane rows of that pitch (0: BAM nibble rows again)
 *   gtx_reads_to_planes    device: the same repack for rows that already lie in HBM (one-off, e.g. behind the upload)
 *   gtx_align_batch_planes gtx_align_batch_flags over plane rows (d_planes 16-byte aligned; d_task_flags may be NULL)
 * gtx_align_batch / gtx_align_batch_flags still take bam_get_seq bytes: they repack into a buffer of the call's scratch
 * first (80 bytes per read more traffic and memory) and run the same kernels. */
int gtx_pack_planes(const uint8_t * seq, uint32_t seq_stride, uint32_t n, uint8_t * planes, uint32_t plane_stride);
int gtx_reads_to_planes(gtx_ctx *, const uint8_t * d_seq, uint32_t seq_stride, uint32_t n_reads, uint8_t * d_planes, uint32_t plane_stride,
                        void * stream);
int gtx_align_batch_planes(gtx_ctx *, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                           uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream);
/* The same, for hosts that keep several batches in flight.  front_event (a hipEvent_t of the caller, may be NULL) is
 * recorded on `stream` behind the position-hinted pass -- the one launch of the call that fills the chip.  What follows are
 * the short queues of the express and general passes (a fraction of a percent of the reads, latency-bound, most CUs idle).
 * tail_stream (may be NULL = `stream`; needs front_event): those passes are launched THERE, behind the event, and the call
 * is complete when tail_stream is -- `stream` is free for the next batch's position-hinted pass, or for the scoring of an
 * earlier one, while the queues drain beside it (bench.py: the staggered schedule, three batches in flight).  Without a
 * position-hinted pass (no hint tables, GTX_HINT=0) the event marks the call's start.
 * done_event (may be NULL): recorded behind the call's LAST launch, on whichever stream that is -- what the reader of the
 * batch's records waits for. */
int gtx_align_batch_planes_staged(gtx_ctx *, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                                  uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream, void * front_event,
                                  void * tail_stream, void * done_event);
/* The same with DENSE RECORDS for what the position-hinted pass finishes.  That pass writes a record of 24 bytes for nearly
 * every read, into slots that lie 2 * rec_words words apart: a 64-byte piece of a line per read for 24 bytes of content, two
 * thirds of what the pass writes (rocprof WRITE_SIZE: 66 bytes per read).  d_compact (16-byte aligned, GTX_COMPACT_WORDS words
 * per read, indexed like d_meta) receives the forward records that fit -- no path or one path without a variant site: every
 * read that adds nothing to the accumulators -- side by side, two store instructions per wavefront over whole lines, and the
 * task's byte of d_task_flags (required) carries GTX_TASK_COMPACT: the record is d_compact[GTX_COMPACT_WORDS * read ...], the
 * task's slot in d_records is NOT written (it keeps what the caller left there).  Records with variant sites or several
 * paths, everything the later passes finish, reverse orientations: in their slots as before, flag clear.  The words of
 * d_compact that belong to other reads are written too (whole lines leave the chip; their content means nothing).
 * Readers: gtx_score_batch_compact; a host that parses records looks at the flag first (graphtyper_amd/lib.py: parse_records). */
#define GTX_TASK_COMPACT 2u
#define GTX_COMPACT_WORDS 8u
int gtx_align_batch_planes_compact(gtx_ctx *, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                                   uint32_t * d_records, uint32_t rec_words, uint32_t * d_compact, uint8_t * d_task_flags, void * stream,
                                   void * front_event, void * tail_stream, void * done_event);

/* Score accumulators (all uint32 / uint64, zero-initialised by the caller; sample-major):
 *   d_log_score [n_samples * total_tri]      HapSample::log_score
 *   d_gt_cov    [n_samples * total_allele]   HapSample::gt_coverage
 *   d_hap_u32   [n_samples * n_hap * 4]      max_log_score, ambiguous_depth, ambiguous_depth_alt, alt_proper_pair_depth
 *   d_stat_u64  [n_hap + 2*total_allele]     per hap mapq_squared; per allele clipped_bp, mapq_squared
 *   d_stat_u32  [n_hap + 6*total_allele]     per hap clipped_reads; per allele score_diff, mismatches, r1f, r1r, r2f, r2r
 *   d_conn_log  [conn_cap * 6]               appended (sample, hap1, allele1, hap2, allele2, count); d_conn_count[0] = entries
 *   d_conn_near [n_samples * total_near]     or NULL.  HapSample::connections between haplotypes less than 100 positions apart
 *                                            (the only ones genotyping reads, hts_parallel_reader.cpp:800-801) as dense counters:
 *                                            haplotype h with the haplotypes h+1 .. near_last[h], entry near_off[h] +
 *                                            allele1 * (alleles of the window) + (allele_off[h2] - allele_off[h+1]) + allele2.
 *                                            With it only the farther pairs (two mates, dense graphs) go to d_conn_log -- a log
 *                                            entry per read and pair does not scale to dense graphs; NULL: everything is logged.
 * Sums are unsaturated; gtx_scores_finalize applies the reference's saturation rules. */
typedef struct gtx_score_buffers
{
  uint32_t n_samples;
  uint32_t * d_log_score;
  uint32_t * d_gt_cov;
  uint32_t * d_hap_u32;
  uint64_t * d_stat_u64;
  uint32_t * d_stat_u32;
  uint32_t * d_conn_log;
  uint32_t * d_conn_count; /* [2]: entries appended, entries dropped because conn_cap was reached */
  uint32_t conn_cap;
  uint32_t * d_conn_near;
  /* SV calling (gtx_params::is_sv_graph), optional: the reference-depth track of ReferenceDepth::add_genotype_paths
   * (src/graph/reference_depth.cpp:109-201) -- per sample the number of accepted reads over every position of the region's
   * reference, which the SV post-processing of the calls reads (reformat_sv_vcf_records).  [n_samples * (ref_depth_len + 1)]
   * uint32, zero-initialised, kept as a DIFFERENCE array (+1 where a read's span starts, -1 behind its end: two atomics
   * per span); gtx_ref_depth_finalize turns the downloaded array into depths.  ref_depth_len = gtx_score_layout::
   * ref_depth_len.  NULL: not kept. */
  uint32_t * d_ref_depth;
  uint32_t ref_depth_len;
} gtx_score_buffers;

int gtx_score_batch(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                    uint32_t rec_words, const gtx_score_buffers * acc, void * stream);
/* ... with the side array of gtx_align_batch_flags (bytes 2 * align_index + orientation relative to d_task_flags, i.e. the
 * array that runs beside d_records); same results */
int gtx_score_batch_flags(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                          const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream);
/* ... and with the items once more in a compact form for the scorer's first stage, which only has to find the few items
 * whose reads carry a variant site (one in six at a SNP per kilobase) and read 40 bytes per item to learn which read an
 * item means: one word per item -- the read's align_index for an item of ONE read aligned forward only
 * (GTX_FLAG_FORWARD_ONLY: what gtx_stream_push makes of an unpaired record, i.e. of most), GTX_ITEM_WORD_FULL for any
 * other item (a pair, a left-over, a read aligned in both orientations: the stage reads the item itself).
 * gtx_item_words (host) fills the array from the items.  Needs d_task_flags; same results. */
#define GTX_ITEM_WORD_FULL 0xFFFFFFFFu
int gtx_item_words(const gtx_score_item * items, uint32_t n_items, uint32_t * words);
int gtx_score_batch_words(gtx_ctx *, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, const uint32_t * d_records,
                          uint32_t rec_words, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream);
/* ... over the records of gtx_align_batch_planes_compact: d_compact and d_task_flags as that call filled them (d_item_words may
 * be NULL: the stage then reads the items); same results */
int gtx_score_batch_compact(gtx_ctx *, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, const uint32_t * d_records,
                            uint32_t rec_words, const uint32_t * d_compact, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream);
/* THE SCORER'S FIRST STAGE BEHIND THE ALIGNMENT (round 6).  update_haplotype_scores_geno (vcf_writer.cpp:88-250) looks at every read;
 * the scorer's first stage finds the few whose records carry a variant site (the side array + the items' words: 5 bytes per
 * item) and leaves their items in a queue for the second stage.  gtx_score_batch_* run it in front of the scoring, on the
 * caller's stream -- for a host that keeps batches in flight (gtx_align_batch_planes_staged) that is the stream of the
 * position-hinted passes, the one that sets the step's time.  gtx_align_batch_planes_triaged is gtx_align_batch_planes_compact
 * (d_compact may be NULL: every record in its slot) that ALSO runs that stage, behind its last alignment launch and on that
 * launch's stream (the tail stream, idle half of the time), into d_work: GTX_WORK_HEADER_WORDS + n_items words, 16-byte aligned,
 * [0] = number of items with work, [GTX_WORK_HEADER_WORDS ...] = their indices (any order).  d_items / d_item_words (the latter
 * may be NULL) as for gtx_score_batch_words; the items must name reads of THIS batch.  done_event is behind the stage.
 * gtx_score_batch_queued scores the items of that queue: the second stage alone; same accumulators as gtx_score_batch_compact
 * over the same records (the order in which items are scored never matters below the saturation guard).  On an SV graph the
 * queue also holds the items that only add to the reference depth.
 * triage_flags: GTX_TRIAGE_ITEMS_ARE_READS -- the caller's promise that item i is read i of this batch, alone and aligned forward
 * only (first.align_index == i, GTX_FLAG_FORWARD_ONLY, no second read: what gtx_stream_push makes of a batch of unpaired
 * records), n_items == n_reads; d_items and d_item_words are not looked at (may be NULL).  The position-hinted pass then leaves
 * one bit per read (a word per wavefront) and the stage reads those, 1/40 of the side bytes and item words, which makes it
 * cheap enough to run beside the next batch's position-hinted pass.  Not for SV graphs (GTX_ERR_UNSUPPORTED). */
#define GTX_WORK_HEADER_WORDS 4u
#define GTX_TRIAGE_ITEMS_ARE_READS 1u
int gtx_align_batch_planes_triaged(gtx_ctx *, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                                   uint32_t * d_records, uint32_t rec_words, uint32_t * d_compact, uint8_t * d_task_flags,
                                   const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, uint32_t triage_flags,
                                   uint32_t * d_work, void * stream, void * front_event, void * tail_stream, void * done_event);
int gtx_score_batch_queued(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                           const uint32_t * d_compact, const uint8_t * d_task_flags, const uint32_t * d_work, const gtx_score_buffers * acc,
                           void * stream);

/* number of score items the kernel refused so far because one read touched more variant sites than its table holds
 * (must be 0 for the accumulators to be complete) */
int gtx_ctx_error_count(gtx_ctx *, uint32_t * out);

/* device pointer and capacity (uint32 words) of the big-record arena; used_words (may be NULL) = words filled since the
 * last rewind, tasks (may be NULL) = (read, orientation) tasks the last gtx_align_batch sent through the second pass
 * (both synchronise with the device) */
int gtx_ctx_big_records(gtx_ctx *, const uint32_t ** d_words, uint64_t * capacity_words, uint64_t * used_words, uint64_t * tasks);

/* (read, orientation) tasks the last gtx_align_batch sent through the exact pass: out[0] with a small part of the slab, out[1]
 * again with a large part, out[2] again with the whole slab, out[3] = tasks that keep a table-overflow status even so
 * (synchronises with the device) */
int gtx_ctx_exact_pass_tasks(gtx_ctx *, uint64_t * out /* [4] */);

/* Durations (ms, HIP events on the launch stream) of the passes of gtx_align_batch -- everything in front of the general
 * pass (position-hinted + express), general, HBM tables -- and the number of tasks handed to the general pass.  The
 * first call only arms the timing.  From then on every call is timed (up to 32 per stream between two queries); a query
 * waits for the calls recorded so far and returns the MEAN duration over them, all streams together -- a host that keeps
 * several calls in flight asks once, behind them -- and the task counts of the last call.  The first timed call after a
 * query starts a new series. */
int gtx_ctx_pass_times(gtx_ctx *, float * ms /* [3] */, uint32_t * queued_for_pass2);

/* The same for the four launches of gtx_align_batch one by one -- position-hinted pass (one read per lane), express pass,
 * general pass, HBM-table pass: ms[4] and the forward tasks each of them completed, tasks[4] (reverse-orientation tasks all
 * go to the general pass and are counted there). */
int gtx_ctx_kernel_times(gtx_ctx *, float * ms /* [4] */, uint32_t * tasks /* [4] */);

/* How many of the 2 * n_reads record slots a gtx_align_batch call filled hold a table-overflow status (GTX_ST_ERROR_MASK in the
 * header's status bits) instead of a result: a full record arena, an exact-pass slab too small, a read longer than the passes
 * take.  Each is a read the accumulators will lack -- a host that wants the reference's result checks this is 0 (gtx_pipeline_run
 * does).  The reverse slots of GTX_FLAG_FORWARD_ONLY reads are never written: they count as what the caller left there (zeros).
 * Synchronises with `stream`. */
int gtx_records_failed(gtx_ctx *, const uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream, uint64_t * out);

/* forget every GTX_ST_EXTERNAL record (call between regions, when d_records is recycled) */
int gtx_ctx_big_records_rewind(gtx_ctx *, void * stream);

/* out[32]: per-phase shader-cycle sums of the alignment kernel; only the profiling build (libgtx_prof.so) fills them */
int gtx_ctx_profile(gtx_ctx *, uint64_t * out);

/* The general pass' task log (profiling build only; the normal build holds none and returns *n = 0): up to `cap` entries of
 * 16 words -- task, workgroup, start cycle, cycles, hardware id, the record's first two words, cycles of phases 0..9 -- into
 * `out`; waits for the device, empties the log. */
int gtx_ctx_profile_log(gtx_ctx *, uint64_t * out, uint64_t cap, uint64_t * n);

/* Per (sample, haplotype) genotype call on the device: replaces get_haplotype_phred (src/typer/vcf.cpp:47-82) and the
 * SampleCall the reference builds from it in Vcf::add_haplotype (src/typer/vcf.cpp:1507-1530; constructor, get_gt_call and
 * get_gq of src/typer/sample_call.cpp:34-131).  Reads the accumulators of gtx_score_batch as they are (unsaturated sums;
 * the clamps of gtx_scores_finalize are applied on the fly).
 *   d_phred [n_samples * total_tri]  uint8   PL of every genotype, layout of d_log_score
 *   d_calls [n_samples * n_hap]      gtx_sample_call, index sample * n_hap + hap */
typedef struct gtx_sample_call
{
  uint16_t gt_first, gt_second;              /* SampleCall::get_gt_call(): first genotype (x <= y) with PL 0 */
  uint16_t ref_total_depth, alt_total_depth; /* SampleCall::ref_total_depth / alt_total_depth */
  uint8_t gq;                                /* SampleCall::get_gq() */
  uint8_t ambiguous_depth, alt_proper_pair_depth;
  uint8_t reserved;
} gtx_sample_call;
int gtx_calls_batch(gtx_ctx *, const gtx_score_buffers * acc, uint8_t * d_phred, gtx_sample_call * d_calls, void * stream);

/* ---- VCF text of the region (host side).  One record per variant site, the line the reference writes
 * for the Variant it makes of a haplotype: Vcf::add_haplotype (src/typer/vcf.cpp:1507-1611) -> Variant::scan_calls /
 * generate_infos (src/typer/variant.cpp:230-1096, VarStats::write_stats src/typer/var_stats.cpp:53-141) ->
 * Vcf::write_record (vcf.cpp:767-1149: CHROM POS ID REF ALT QUAL FILTER INFO GT:AD:MD:DP:GQ:PL, PL and GQ through
 * binned_pl) with the region filter of Vcf::write_records (vcf.cpp:1161-1275).  In front of the records stands the column
 * line (#CHROM ... FORMAT sample names); the description lines of the header are not produced.
 * All arrays are HOST copies: the accumulators of gtx_score_batch (raw sums or finalized) and the outputs of
 * gtx_calls_batch for the same n_samples.  Not built: variant break-down / pool merge (vcf_operations.cpp).
 * SV graphs (`genotype_sv`): the sites go through the SV post-processing of the calls instead -- reformat_sv_vcf_records
 * (src/graph/sv.cpp:117-655: one bi-allelic record per SV allele at the SV's own position, REF N, ALT <TYPE:SVSIZE=n:MODEL>,
 * models BREAKPOINT(1/2), COVERAGE from the reference-depth track and AGGREGATED), the sort and stats.clear() of the pool's
 * writer (src/utilities/hts_parallel_reader.cpp:1003-1020) and what vcf_merge_and_break does for genotype_sv
 * (force_no_break_down: normalize, generate_infos, drop a record nobody was called with; vcf_operations.cpp:480-700), written
 * in the order and with the ID suffixes of Vcf::write_records (vcf.cpp:1161-1275); FORMAT GT:FT:AD:MD:DP:RA:PP:GQ:PL.
 * Needs sv_table and ref_depth.  GTX_ERR_UNSUPPORTED: a site that mixes SV and non-SV alleles, a breakend allele that starts
 * with its tag, an SV of 40 bases or fewer where the coverage model is asked.
 * Writes min(*len, cap) bytes to out (may be NULL with cap 0 to ask for the length).  Large jobs (sites x samples >=
 * 200 000) are written by a team of host threads over ranges of sites (GTX_HOST_THREADS, default up to 32); the text does
 * not depend on the team. */
typedef struct gtx_vcf_request
{
  const char * contig;               /* CHROM */
  const char * const * sample_names; /* [n_samples] */
  uint32_t n_samples;
  uint32_t region_begin, region_end; /* 1-based positions on the contig, inclusive; sites outside are skipped */
  int32_t filter_zero_qual;          /* FILTER_ZERO_QUAL of write_record: sites with QUAL 0 are skipped */
  const char * variant_suffix_id;    /* Options::variant_suffix_id or NULL */
  const uint32_t * gt_cov;           /* [n_samples * total_allele] */
  const uint64_t * stat_u64;         /* layout of gtx_score_buffers::d_stat_u64 */
  const uint32_t * stat_u32;         /* layout of gtx_score_buffers::d_stat_u32 */
  const uint8_t * phred;             /* [n_samples * total_tri] of gtx_calls_batch */
  const gtx_sample_call * calls;     /* [n_samples * n_hap] of gtx_calls_batch */
  /* SV graphs only (gtx_params::is_sv_graph; else ignored, may be NULL / 0): */
  const char * sv_table;             /* Graph::SVs as text: gtx_graph_sv_table of the graph the context was made from */
  const uint32_t * ref_depth;        /* [n_samples * (ref_depth_len + 1)]: the downloaded d_ref_depth after gtx_ref_depth_finalize */
  uint32_t ref_depth_len;            /* gtx_score_layout::ref_depth_len */
} gtx_vcf_request;
int gtx_vcf_records(const gtx_ctx *, const gtx_vcf_request *, char * out, uint64_t cap, uint64_t * len);

/* The records of the FINAL file of a small-variant graph: replaces vcf_merge_and_break with the break-down
 * (src/typer/vcf_operations.cpp:480-732, force_no_break_down = false; genotype() ends with it, src/utilities/genotype.cpp:577-604).
 * Every site goes through break_down_variant (src/typer/variant.cpp:1652-1713): alleles of one length are taken apart position by
 * position (break_multi_snps, :1996-2111 -- a base in front first when the alleles do not start alike; only alleles somebody is
 * called with make a SNP, the calls keep the smallest PL of the genotypes that fall together, depths and read statistics are
 * added: update_per_allele_stats, :34-82), so a site nobody carries an alternative allele of leaves no record; what comes out is
 * normalised (Variant::normalize, :1256-1315), judged (generate_infos: a record whose every alternative allele is bad is dropped
 * unless no_filter_bad_alts) and written by Vcf::write_records in the reference's windows.  Alleles of DIFFERENT lengths go
 * through paw::Skyr in the reference (break_down_skyr, :2113-2190), a library its tree does not hold: with no_variant_overlapping
 * (the reference's --no_variant_overlapping, and the second file of --normal_and_no_variant_overlapping) such a site is written
 * whole, as the reference writes it then; without it the call returns GTX_ERR_UNSUPPORTED when the graph has such a site that
 * SOMEBODY IS CALLED WITH AN ALTERNATIVE ALLELE OF (a site nobody carries is handed to paw::Skyr as the reference allele throughout,
 * :2137-2155, and leaves no record: that case is made).  A host that wants the reference's default file for a region with carried
 * indels has to take the no_variant_overlapping one and decompose those sites itself.
 * Graphs of SNPs (cfg2) have none: both modes are the reference's file.  First line: the column line.  Not for SV graphs. */
int gtx_vcf_records_final(const gtx_ctx *, const gtx_vcf_request *, int no_variant_overlapping, int no_filter_bad_alts, char * out, uint64_t cap,
                          uint64_t * len);

/* The header in front of the records: replaces Vcf::write_header (src/typer/vcf.cpp:526-760) -- ##fileformat, ##fileDate,
 * ##source, the version / branch / SHA1 lines (the reference prints its build's constants: given here), one ##contig line per
 * contig (Graph::contigs), the ##INFO / ##FORMAT / ##FILTER description lines and the column line (with FORMAT and the sample
 * names unless genotypes are dropped or there is no sample).  gtx_vcf_records' own first line is that column line too: a
 * caller that writes a file takes the header from here and the records from there without their first line. */
typedef struct gtx_vcf_header_request
{
  const char * file_date;  /* YYYYMMDD (current_date(), vcf.cpp:36-45) */
  const char * version;    /* graphtyper_VERSION_MAJOR.MINOR.PATCH */
  int32_t dirty;           /* GIT_NUM_DIRTY_LINES != 0: "-dirty" behind the version */
  const char * git_branch; /* GIT_BRANCH */
  const char * git_sha1;   /* GIT_COMMIT_LONG_HASH */
  const char * const * contig_names;
  const uint32_t * contig_lengths;
  uint32_t n_contigs;
  const char * const * sample_names;
  uint32_t n_samples;
  int32_t drop_genotypes; /* is_dropping_genotypes */
} gtx_vcf_header_request;
int gtx_vcf_header(const gtx_vcf_header_request *, char * out, uint64_t cap, uint64_t * len);

/* BGZF members of `in` (SAM spec 4.1: gzip members of at most 0xff00 input bytes with the BC extra field), with_eof: followed
 * by the 28-byte empty member that ends a file -- what the reference's bgzf_stream (include/graphtyper/utilities/
 * bgzf_stream.hpp) makes of the text it is given.  level: zlib's 0..9, -1 = default.  out may be NULL with cap 0 to ask for
 * the size.  The members of an input of a megabyte and more are deflated on up to eight threads (the bytes are the same). */
int gtx_bgzf_compress(const void * in, uint64_t in_len, int level, int with_eof, void * out, uint64_t cap, uint64_t * out_len);
/* The other direction for ONE member's payload: the raw DEFLATE stream `in` (RFC 1951; what lies between a BGZF member's
 * header and its CRC32) into exactly out_len bytes -- the decoder the BAM readers use (graphtyper_amd/csrc/gtx_inflate.hpp:
 * whole members of known size, 64-bit bit buffer, one table look-up per symbol; the reference reads through htslib's
 * bgzf.c, which calls libdeflate or zlib).  GTX_ERR_IO: not a valid stream of that size. */
int gtx_inflate_raw(const void * in, uint64_t in_len, void * out, uint64_t out_len);

/* Coordinate index of a bgzip-compressed VCF file.  gtx_tabix_build replaces Vcf::write_tbi_index (src/typer/vcf.cpp:1308-1321:
 * htslib's tbx_index_build with the VCF preset): min_shift 0 writes <vcf>.tbi (bins of 16 kb .. 512 Mb and the linear index),
 * min_shift > 0 a .csi of that geometry (the reference's --csi uses 14) -- to index_path when given.  The file has to be
 * BGZF (gtx_bgzf_compress, bgzip) and sorted by contig and position.  The formats are those of the tabix and CSI
 * specifications; htslib's optional merging of sparse bins is not reproduced (any reader of the formats accepts the result).
 * gtx_tabix_start: the virtual offset from which a sequential read finds every record of `chrom` that overlaps
 * [begin, end) (0-based, end exclusive), from the .tbi / .csi beside the file; *any = 0: the index knows of none.
 * gtx_graph_from_files uses the same look-up when its VCF has an index (the reference reads a region's records through
 * tabix, src/graph/constructor.cpp:163-176). */
int gtx_tabix_build(const char * vcf_gz_path, int min_shift, const char * index_path /* NULL: <vcf>.tbi or <vcf>.csi */);
int gtx_tabix_start(const char * vcf_gz_path, const char * chrom, int64_t begin, int64_t end, uint64_t * voffset, int * any);

/* ---- multi-GPU: reads shard over the GPUs of a node (one process per GPU, graph + index replicated), the accumulators
 * are summed once per region.  The reference's counterpart is the merge of per-thread / per-pool results
 * on the host (src/typer/caller.cpp:439-482, src/typer/vcf_operations.cpp:366-374); every per-read effect is an integer
 * addition, so sum-then-clamp (gtx_scores_finalize) equals one sequential pass below the saturation guard.
 *
 * gtx_scores_alloc puts all accumulators of gtx_score_buffers into ONE device block, zeroed:
 *   [stat_u64][log_score][gt_cov][hap_u32][stat_u32][conn_near][ref_depth (SV graphs)] [conn_count][conn_log]
 * The first *reduced_bytes (may be NULL) are what gtx_scores_reduce sums; the connection log stays rank-local (far pairs:
 * whoever reads them concatenates the ranks' logs).  gtx_scores_zero clears the block for the next region (async on
 * `stream`), gtx_scores_free releases it.
 * gtx_scores_reduce: in-place sum over the ranks of `rccl_comm` (an ncclComm_t) on `stream`: one RCCL group -- the u64
 * statistics and the u32 counters as two all-reduce operations fused into one launch (u64 sums cannot travel as pairs of
 * u32).  Buffers that were not made by gtx_scores_alloc are accepted (one operation per array, same group).
 * RCCL is bound at run time (the process' own librccl.so when it has one); without it these calls return
 * GTX_ERR_UNSUPPORTED.  gtx_comm_*: thin helpers for hosts that have no communicator yet -- rank 0 makes the id
 * (ncclGetUniqueId), sends its GTX_COMM_ID_BYTES bytes to the other ranks by any means, every rank calls
 * gtx_comm_init_rank (ncclCommInitRank on `device`). */
#define GTX_COMM_ID_BYTES 128
int gtx_scores_alloc(gtx_ctx *, uint32_t n_samples, uint32_t conn_cap, gtx_score_buffers * out, uint64_t * reduced_bytes);
int gtx_scores_zero(gtx_ctx *, const gtx_score_buffers *, void * stream);
int gtx_scores_free(gtx_ctx *, gtx_score_buffers *);
int gtx_scores_reduce(gtx_ctx *, const gtx_score_buffers *, void * rccl_comm, void * stream);
int gtx_comm_unique_id(void * id /* [GTX_COMM_ID_BYTES] */);
int gtx_comm_init_rank(const void * id, int n_ranks, int rank, int device, void ** comm);
int gtx_comm_destroy(void * comm);

/* The one order-dependent step of the scoring: Haplotype::explain_to_score stops adding to a (haplotype, sample) once its
 * max_log_score has come within epsilon of 0xFFFF (src/graph/haplotype.cpp:560; roughly 8 000 reads over one site in one
 * sample), and which reads are dropped then depends on the order of the calls.  gtx_score_batch adds without the guard.
 * gtx_scores_replay finds the cells whose sum reached the guard, goes over the region's items once more -- ALL of them, in
 * the order they were produced (gtx_stream_push / gtx_stream_finish order = the order in which genotype_only reaches the
 * VcfWriter), against the same resident d_records -- logs the explain_to_score calls on those cells, replays them one by
 * one on the host as the reference does and stores the exact log_score rows and max_log_score back into `acc` (device).
 * Call it after the last gtx_score_batch of a region (single process: with reads sharded over several GPUs the call order
 * is spread over the ranks) and before gtx_calls_batch / downloading.  n_replayed: cells replayed; n_unsupported: cells at
 * the guard that lie on a site of more than 64 alleles (left as they are, reported by gtx_scores_finalize).  Synchronises
 * with `stream`. */
int gtx_scores_replay(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                      const gtx_score_buffers * acc, void * stream, uint64_t * n_replayed, uint64_t * n_unsupported);
/* ... over the records of gtx_align_batch_planes_compact (a pair's mate without a variant site has its record there) */
int gtx_scores_replay_compact(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                              const uint32_t * d_compact, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream,
                              uint64_t * n_replayed, uint64_t * n_unsupported);

/* The same for reads sharded over several GPUs ("detect the overflow case and replay that (haplotype, sample)
 * sequentially").  After gtx_scores_reduce every rank's block holds the sums over all ranks, so every rank finds the same cells
 * at the guard.  gtx_scores_replay_log: this rank's items once more against that block -- the explain_to_score calls on those
 * cells, `item` = item_base + the item's index (item_base: where this rank's items stand in the region's sequence, i.e. the
 * reference's call order over all ranks); *n entries into `out` (GTX_ERR_CAPACITY with *n = the number wanted when cap is too
 * small); d_compact / d_task_flags: NULL, or the dense records of gtx_align_batch_planes_compact.  The hosts exchange the logs
 * (they are plain data: an all-gather of bytes), and gtx_scores_replay_apply -- on the rank(s) that go on to gtx_calls_batch --
 * replays the entries of ALL ranks in call order and stores the exact rows into its block, as gtx_scores_replay does for one
 * process.  Both synchronise with `stream`. */
typedef struct gtx_replay_entry
{
  uint32_t item, cell;       /* place in the region's item sequence; sample * n_hap + haplotype */
  uint32_t order_eps;        /* epsilon | which read of the item << 8 */
  uint32_t mask_lo, mask_hi; /* the alleles the read explains */
  uint32_t pad;
} gtx_replay_entry;
int gtx_scores_replay_log(gtx_ctx *, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                          const uint32_t * d_compact, const uint8_t * d_task_flags, const gtx_score_buffers * acc, uint32_t item_base, void * stream,
                          gtx_replay_entry * out, uint64_t cap, uint64_t * n, uint64_t * n_unsupported);
int gtx_scores_replay_apply(gtx_ctx *, const gtx_score_buffers * acc, const gtx_replay_entry * entries, uint64_t n_entries, void * stream,
                            uint64_t * n_replayed);

/* Host-side clamp of downloaded accumulators to the reference's stored types (haplotype.cpp:19-44: u8 -> 255,
 * u16 -> 0xFFFF).  Returns the number of (haplotype,sample) cells whose max_log_score reached the sequential
 * saturation guard of explain_to_score (haplotype.cpp:560) -- those cells need a sequential replay and are
 * reported, never silently clamped. */
int gtx_scores_finalize(uint32_t * log_score, uint64_t n_log, uint32_t * gt_cov, uint64_t n_cov, uint32_t * hap_u32,
                        uint64_t n_hap_cells, uint64_t * n_saturated);

/* In-place on the downloaded difference array of gtx_score_buffers::d_ref_depth: running sums per sample, clamped to the
 * reference's uint16 (the first ref_depth_len words of every sample's row become its depths; the last word is scratch).
 * *n_saturated (may be NULL) counts the positions that reached 0xFFFF.  The reference saturates there when a read has
 * several paths and wraps around when it has one (reference_depth.cpp:141-145 has no check); this function clamps. */
int gtx_ref_depth_finalize(uint32_t * ref_depth, uint32_t n_samples, uint32_t ref_depth_len, uint64_t * n_saturated);

/* Phasing flags between alt alleles of variant sites less than 100 bp apart: replaces the `ph` construction of
 * parallel_reader_genotype_only (src/utilities/hts_parallel_reader.cpp:782-904).  Host only.
 * gt_cov: finalised d_gt_cov; conn_log / n_conn: the downloaded connection log; conn_near: the downloaded d_conn_near (or
 * NULL when the batches were scored without it).  Rows come in the order of the reference's
 * std::map (hap1, allele1, hap2, allele2); an outer key that exists without any flag under it (the reference inserts it
 * as soon as a connection exists) is one row with hap2 = allele2 = 0xFFFF, flags = 0.
 * flags: 1 = IS_ANY_HAP_SUPPORT, 2 = IS_ANY_ANTI_HAP_SUPPORT (include/graphtyper/constants.hpp.in:56-57), or-ed over samples.
 * *n receives the number of rows (GTX_ERR_CAPACITY when it exceeds cap). */
typedef struct gtx_phase_entry
{
  uint16_t hap1, allele1, hap2, allele2;
  int8_t flags;
  uint8_t reserved;
} gtx_phase_entry;
int gtx_phase_flags(const gtx_ctx *, uint32_t n_samples, const uint32_t * gt_cov, const uint32_t * conn_log, uint64_t n_conn,
                    const uint32_t * conn_near, /* the downloaded d_conn_near or NULL */
                    gtx_phase_entry * out, uint64_t cap, uint64_t * n);

/* The sites one genotyping iteration hands to the next: replaces vcf_merge_and_filter (src/typer/vcf_operations.cpp:278-478; the
 * file genotype() feeds to the next construct_graph, src/utilities/genotype.cpp:520-575).  Every alternative allele that
 * Variant::generate_infos keeps (variant.cpp:1040-1070: reads reached it, QD per allele >= 1, best support in one sample >= 2,
 * stricter on sites of 71 / 131 and more alleles) is a bi-allelic record of its own without samples -- QUAL 0, FILTER ".", INFO
 * GT_ID (the allele's number over the file, from 1, dropped alleles counted), GT_ANTI_HAPLOTYPE (later kept alleles of its site,
 * then the alleles `ph` flags IS_ANY_ANTI_HAP_SUPPORT and nothing else), GT_HAPLOTYPE (those flagged IS_ANY_HAP_SUPPORT and
 * nothing else).  rq: as for gtx_vcf_records over the accumulators of ALL pools (the reference adds the pools' statistics;
 * here they are one block); region / filter_zero_qual / sample names are not used: every site of the graph is judged.
 * ph / n_ph: the rows of gtx_phase_flags, in its order.  First line: the column line.  Not for SV graphs (GTX_ERR_UNSUPPORTED). */
int gtx_vcf_sites(const gtx_ctx *, const gtx_vcf_request *, const gtx_phase_entry * ph, uint64_t n_ph, char * out, uint64_t cap, uint64_t * len);

/* ---- host mirror of the per-record control flow (no device work) ----
 * Feed records in merged stream order; the stream decides which records are filtered, which reuse the previous
 * alignment (equal_pos_seq) and which pairs / unpaired reads reach the scorer. */
typedef struct gtx_stream gtx_stream;

typedef struct gtx_stream_record
{
  uint16_t flag;
  uint8_t mapq;
  uint8_t score_diff;
  int32_t tid, mtid;
  int32_t pos;
  int32_t isize;
  uint16_t l_qseq;
  uint16_t rg;      /* read group index (mate maps are per read group) */
  uint32_t sample;  /* pn_index */
  uint64_t name_id; /* identity of the read name (equal ids <=> equal QNAME) */
  /* only read when gtx_params::is_sv_graph (record filter of SV calling, hts_parallel_reader.cpp:528-568) */
  int32_t mpos;
  uint32_t n_cigar;
  uint32_t cigar_front, cigar_back; /* raw BAM cigar words (op | len << 4) of the first / last operation */
} gtx_stream_record;

int gtx_stream_create(const gtx_params * params, uint32_t n_read_groups, gtx_stream ** out);
void gtx_stream_destroy(gtx_stream *);
/* seq: n * seq_stride bytes of BAM 4-bit packed bases (host).  For every record appends at most one alignment task
 * (copying its packed sequence + gtx_read_meta) and at most one score item.  Capacities are in elements. */
int gtx_stream_push(gtx_stream *, const gtx_stream_record * recs, const uint8_t * seq, uint32_t seq_stride, uint32_t n,
                    uint8_t * align_seq, gtx_read_meta * align_meta, uint32_t align_cap, uint32_t * n_align,
                    gtx_score_item * items, uint32_t item_cap, uint32_t * n_items);
/* plane_stride != 0: align_seq receives plane rows of that pitch (see gtx_pack_planes) instead of copies of the BAM bytes */
int gtx_stream_set_planes(gtx_stream *, uint32_t plane_stride);
/* SV calling only, optional: the (extreme) coverage filter (hts_parallel_reader.cpp:594-633) drops a record once its
 * sample has more than avg_cov_by_readlen[sample] * 150 accepted records in the record's 50 bp bin.  Without this call
 * (or with a value <= 0 for a sample) nothing is dropped -- Options::no_filter_on_coverage. */
int gtx_stream_set_coverage(gtx_stream *, const double * avg_cov_by_readlen, uint32_t n_samples);
/* End of the record stream.  SV calling: emits one GTX_ITEM_LEFTOVER item per read still waiting for its mate
 * (hts_parallel_reader.cpp:717-772); always forgets the parked reads. */
int gtx_stream_finish(gtx_stream *, gtx_score_item * items, uint32_t item_cap, uint32_t * n_items);
/* number of accepted / duplicated records so far and mates still parked */
int gtx_stream_counts(const gtx_stream *, uint64_t * n_records, uint64_t * n_duplicated, uint64_t * n_parked);

/* ---- BAM ingest in front of gtx_stream_push (host).  Replaces, for BAM files, what the reference
 * does with htslib before a record reaches genotype_only(): HtsReader::open (src/utilities/hts_reader.cpp:17-124: header,
 * @RG lines -> read group and sample tables, sample name from the file name when there is none),
 * HtsReader::get_next_read_in_order (:166-303: the records of one position sorted by length, then packed bases),
 * HtsParallelReader::open / read_record (src/utilities/hts_parallel_reader.cpp:66-136: k-way merge of the files by
 * (tid, pos, l_qseq, packed bases), include/graphtyper/utilities/hts_utils.hpp:48-108), get_sample_and_rg_index
 * (hts_reader.cpp:354-387) and get_score_diff (src/typer/alignment.cpp:140-325).  Records come out as gtx_stream_record +
 * packed bases, ready for gtx_stream_push; samples and read groups are numbered across the files in the order given.
 * Needs zlib only (BGZF members are inflated one by one, so virtual offsets can be sought).  `region` ("chr",
 * "chr:begin-end", 1-based inclusive; NULL, "" or "." = everything) yields the records that overlap it, as sam_itr_querys
 * would return them: with a .bai beside the file (<bam>.bai or <name>.bai) the scan starts at the first place the index
 * allows an overlapping record -- or a .csi (<bam>.csi) --, without one at the head of the file.  Not read: CRAM.  Records with equal sort keys keep file order (the reference's
 * std::sort / heap leave the order of exact duplicates open; results do not depend on it).
 * gtx_reads_next fills up to cap records (n = 0: end); a read whose packed bases exceed seq_stride is an error.
 * Threads: a gtx_reads is used by one thread at a time; several may be open on several host threads.  While any is open
 * the library keeps a team of worker threads that inflate BGZF members ahead of the readers (environment:
 * GTX_BGZF_THREADS, default up to 16, 0 = none: every reader inflates its own members); it is gone when the last closes. */
typedef struct gtx_reads gtx_reads;
int gtx_reads_open(const char * const * bam_paths, uint32_t n_paths, const char * region, gtx_reads ** out);
int gtx_reads_info(const gtx_reads *, uint32_t * n_samples, uint32_t * n_read_groups);
const char * gtx_reads_sample_name(const gtx_reads *, uint32_t i);
int gtx_reads_next(gtx_reads *, gtx_stream_record * recs, uint8_t * seq, uint32_t seq_stride, uint32_t cap, uint32_t * n);
void gtx_reads_close(gtx_reads *);

/* ---- the host loop around the path, inside the library.  gtx_pipeline_run replaces the reference's worker threads over BAM
 * pools (src/typer/caller.cpp:399-436, each running parallel_reader_genotype_only, src/utilities/hts_parallel_reader.cpp:
 * 245-338): n_threads host threads, thread k with the files k, k + n_threads, ... (merged like gtx_reads_open merges them),
 * each running gtx_reads_next -> gtx_stream_push (plane rows) -> pinned staging -> H2D -> gtx_align_batch_planes ->
 * gtx_score_batch_flags on a stream of its own, two staging sets deep, all into the ONE accumulator block `acc` of the
 * context (made by gtx_scores_alloc for at least the files' samples; samples are numbered by name in the order the groups
 * bring them: position-sliced files of one sample are one sample).  chunk: records per batch; rec_words: words of a record slot; record_slots_per_thread: how many reads a thread's
 * files may hold at most (their records stay on the device for the run: a mate's item names a task of batches ago).
 * What follows is the caller's: gtx_calls_batch, gtx_vcf_records.  Reads of more than 160 bases are not taken by this loop. */
typedef struct gtx_pipeline_stats
{
  uint64_t records, tasks, items;           /* records read; alignment tasks and score items made of them */
  double decode_s, push_s, enqueue_s;       /* summed over the threads: gtx_reads_next, gtx_stream_push, copies + launches */
  double slowest_thread_s;                  /* the largest per-thread sum of the three */
  double loop_s;                            /* from the moment every thread has its buffers to the last stream's end */
  double wall_s;                            /* the whole call (opening the files and allocating included) */
  uint32_t n_samples, n_threads;
  /* what a capacity limit dropped -- any of them non-zero and the call returns GTX_ERR_CAPACITY (the block is then not a result): */
  uint64_t records_failed;                  /* record slots with a table-overflow status (gtx_records_failed over every thread's slots) */
  uint64_t score_items_refused;             /* gtx_ctx_error_count's increase over the run */
  uint64_t connections_dropped;             /* far-pair connections beyond the block's log (conn_cap of gtx_scores_alloc) */
} gtx_pipeline_stats;
/* A non-OK return leaves `acc` undefined (the threads that did not fail have added into it): gtx_scores_zero before it is used again. */
int gtx_pipeline_run(gtx_ctx *, const char * const * bam_paths, uint32_t n_paths, uint32_t n_threads, const char * region, uint32_t chunk,
                     uint32_t rec_words, uint64_t record_slots_per_thread, const gtx_score_buffers * acc, gtx_pipeline_stats * stats);

/* ---- region after region, inside the library.  gtx_regions_run replaces the loop of genotype_regions
 * (src/utilities/genotype.cpp:735-738, "Genotype regions serially": every region a call of genotype(), :406-604) for the last
 * iteration of each region -- the graph made of the variant records the iterations before agreed on, the reads genotyped on it,
 * the region's VCF records written: per job gtx_graph_build -> gtx_ctx_create -> gtx_align_batch_planes ->
 * gtx_score_batch_flags -> gtx_calls_batch -> gtx_vcf_records.  The reference keeps its threads inside one region; a 50 kb
 * region is a third of a millisecond of device work here and three times that of host work around it, so the library overlaps
 * the stages of DIFFERENT regions: n_builders host threads make graphs and contexts ahead (0 = 4), n_device_threads threads
 * with a stream each run the reads (0 = 2), n_text_threads write the text (0 = 3).  A job's text depends on that job's inputs
 * only -- the same bytes as the six calls made one after the other.
 * Per job, in: the region's reference and records as for gtx_graph_build; the reads RESIDENT on `device` as plane rows
 * (gtx_reads_to_planes / gtx_stream_push) with their gtx_read_meta, the score items (gtx_stream_push's, align_index counted
 * from the job's first read); vcf_begin / vcf_end / filter_zero_qual as in gtx_vcf_request.  Out: text (column line first;
 * malloc'ed by the library, released by gtx_regions_free), text_len, status.
 * rec_words: words of a record slot (>= 8); conn_cap: as for gtx_scores_alloc.  A job whose records, score items or connections
 * ran into a capacity limit fails with GTX_ERR_CAPACITY (no text); the call returns the first failing job's status and goes
 * on with the others.  Small-variant graphs (params->is_sv_graph = 0). */
typedef struct gtx_region_job
{
  const char * reference;            /* [reference_len] the region's bases, reference[0] at contig position region_begin */
  uint64_t reference_len;
  int64_t region_begin, region_end;  /* 0-based, [begin, end) */
  const gtx_record * records;        /* sorted by pos */
  uint32_t n_records;
  int32_t add_all_variants;
  const uint8_t * d_planes;          /* device: [n_reads] plane rows */
  uint32_t plane_stride;
  const gtx_read_meta * d_meta;      /* device: [n_reads] */
  uint64_t n_reads;
  const gtx_score_item * d_items;    /* device: [n_items] */
  uint64_t n_items;
  uint32_t vcf_begin, vcf_end;       /* gtx_vcf_request::region_begin / region_end (1-based, inclusive) */
  int32_t filter_zero_qual;
  int32_t status;                    /* out */
  char * text;                       /* out */
  uint64_t text_len;                 /* out */
} gtx_region_job;
typedef struct gtx_regions_stats
{
  double graph_build_s, ctx_create_s, device_s, vcf_text_s; /* summed over the threads of each stage */
  double wall_s;
  uint32_t n_builders, n_device_threads, n_text_threads;
  uint32_t reserved;
  uint64_t records_failed, score_items_refused, connections_dropped; /* summed over the jobs that failed with GTX_ERR_CAPACITY */
} gtx_regions_stats;
int gtx_regions_run(gtx_region_job * jobs, uint32_t n_jobs, const gtx_params * params, int device, const char * contig,
                    const char * const * sample_names, uint32_t n_samples, uint32_t rec_words, uint32_t conn_cap,
                    uint32_t n_builders, uint32_t n_device_threads, uint32_t n_text_threads, gtx_regions_stats * stats);
void gtx_regions_free(gtx_region_job * jobs, uint32_t n_jobs);

/* ---- the read pre-filter in front of the ingest (host).  gtx_bam_shrink replaces gyper::bamshrink / bamshrink_multi
 * (src/utilities/bamshrink.cpp:1248-1371; the work is qualityFilterSlice2, :667-1045): from a coordinate-sorted BAM file it
 * writes a BAM file with the records around the intervals that the caller is going to look at -- pairs and single reads that
 * pass the mapping-quality / clipping / matching-bases / base-quality filters, adapters cut off pairs whose fragment is not
 * longer than a read, reads dropped by their AS / XS / WS scores, Ns cut off the ends, at most max_bin_sum pairs per 50
 * positions, only the RG / AS / XS / WS tags, two-level qualities, short read names -- sorted by begin position, in the
 * reference's order.  Intervals are the reference's (chrom, begin, end): 0-based, both inside; records are taken from
 * max_frag_len - 100 positions around each.  With ONE interval the output header keeps the @HD / @RG lines and that contig
 * only and the records get contig number 0 (:909-922, :1304-1335); with several the header is copied.  The index beside the
 * file (.bai / .csi, as for gtx_reads_open) is used when there is one; without one the file is scanned (the reference
 * refuses).  gtx_shrink_params_default fills in the reference's option defaults (include/graphtyper/utilities/
 * options.hpp:50,63-69,90); avg_cov_by_readlen <= 0 means "unknown": the default bin cap, and no record dropped as
 * "super high depth" (:1268-1271).  The output is read back with gtx_reads_open.  Not read: CRAM. */
typedef struct gtx_shrink_params
{
  int32_t max_frag_len;           /* bamshrink_max_fraglen, 1000 */
  int32_t min_num_matching;       /* bamshrink_min_matching, 55 */
  int32_t filter_mapq0;           /* !bamshrink_is_not_filtering_mapq0, 1 */
  int32_t no_filter_on_coverage;  /* 0 */
  int32_t min_read_len;           /* bamshrink_min_readlen, 75 */
  int32_t min_read_len_low_mapq;  /* bamshrink_min_readlen_low_mapq, 94 */
  int32_t min_unpaired_read_len;  /* bamshrink_min_unpair_readlen, 94 */
  int32_t sam_flag_filter;        /* 3840 */
  int64_t as_filter_threshold;    /* bamshrink_as_filter_threshold, 40 */
  double avg_cov_by_readlen;      /* the sample's average coverage / read length; <= 0: unknown */
  int32_t change_read_names;      /* 1: names become short counters (what a release build of the reference does) */
  int32_t compress_level;         /* BGZF level of the output, 1 */
} gtx_shrink_params;
typedef struct gtx_shrink_stats
{
  uint64_t records_read;      /* records of the intervals' surroundings */
  uint64_t records_written;
  uint64_t pairs_kept;        /* pairs that went to the output queue */
  uint64_t singles_kept;      /* single reads that did */
  uint64_t dropped_by_depth;  /* reads dropped because their bin was full */
} gtx_shrink_stats;
void gtx_shrink_params_default(gtx_shrink_params *);
int gtx_bam_shrink(const char * bam_in, const char * const * chroms, const int32_t * begins, const int32_t * ends, uint32_t n_intervals,
                   const gtx_shrink_params * params /* NULL: defaults */, const char * bam_out, gtx_shrink_stats * stats /* may be NULL */);
/* bamshrink_multi (src/utilities/bamshrink.cpp:1352-1371): the intervals come from a file of "contig first last" lines (1-based,
 * sorted); neighbours that begin within 2 * max_frag_len of the one before are one interval (readIntervals, :1047-1130). */
int gtx_bam_shrink_multi(const char * bam_in, const char * interval_file, const gtx_shrink_params * params, const char * bam_out,
                         gtx_shrink_stats * stats);

/* ---- variant discovery, first slice.  Replaces, per sample, the first pass over the reads of a region:
 * run_first_pass (src/typer/caller.cpp:488-1186) -- the SNP and indel events its CIGAR walk reads off the alignments
 * (caller.cpp:583-775, Event / EventSupport include/graphtyper/typer/event.hpp:30-113, add_snp_event_to_bucket /
 * add_indel_event_to_bucket src/typer/bucket.cpp:75-182), the correction for reads with 12 and more events and the phase
 * counts between the events of a read (caller.cpp:777-822), the coverage difference arrays, and the two support filters
 * (EventSupport::has_good_support src/typer/event.cpp:226-256 for SNPs; good / realignment support of indels,
 * caller.cpp:990-1186).  Not built: what follows -- the haplotypes of the surviving events, the realignment of reads to the
 * indels (paw::pairwise_alignment, a dependency that is absent from the reference tree), the second pass, the pool merge.
 *   gtx_disc_create        the region's reference (upper-case letters; reference[0] = contig position region_begin, 0-based) on `device`
 *   gtx_disc_events_batch  device: the events of n_reads reads in stream order.  d_planes: the reads as plane rows (gtx_pack_planes),
 *                          d_qual: their base qualities (qual_stride bytes per read), d_reads / d_cigar: the bam1_t fields and the raw
 *                          BAM cigar words.  d_events receives the events, a read's own behind each other in CIGAR order;
 *                          d_read_out[i] says where (first_event, n_events).  d_counts[0] += events produced, d_counts[1] += events
 *                          that did not fit event_cap (both zeroed by the caller; a pass with d_counts[1] != 0 has to be repeated
 *                          with a larger buffer).
 *   gtx_disc_first_pass    host: the downloaded arrays -> the events that survive the pass with their support, as a word stream:
 *                          per event pos, type, length, its characters; hq_count, lq_count, proper_pairs, first_in_pairs,
 *                          sequence_reversed, clipped, max_mapq, max_distance, uniq_pos1..3, span, has_realignment_support,
 *                          has_indel_good_support, max_log_qual, number of phase entries; then per phase entry the other event
 *                          (pos, type, length, characters) and its count.  Events in the reference's order (position; insertions,
 *                          deletions, SNPs; sequence).  seq: the reads' BAM nibble rows (the inserted bases of an insertion are read
 *                          from them), bucket_size: BUCKET_SIZE of the reference (it decides nothing about the result). */
typedef struct gtx_disc gtx_disc;
typedef struct gtx_disc_read
{
  int32_t pos;       /* core.pos */
  uint16_t flag;     /* core.flag */
  uint8_t mapq;      /* core.qual */
  uint8_t reserved;
  uint16_t l_qseq;
  uint16_t n_cigar;
  uint32_t cigar_off; /* its first word in the cigar array */
} gtx_disc_read;
typedef struct gtx_disc_event
{
  uint32_t read;         /* index of the read in its batch */
  uint32_t pos;          /* Event::pos: contig position, 0-based */
  uint32_t seq;          /* 'X': the read's base (ASCII); 'I': offset of the first inserted base in the read; 'D': offset of the first deleted base in the region */
  uint16_t len;          /* Event::sequence.size() */
  uint8_t type;          /* 'X', 'I', 'D' */
  uint8_t hq;            /* 'X': base quality >= 25; indels: 1 */
  uint16_t max_distance; /* 'X': min(read_pos, l_qseq - 1 - read_pos) */
  uint16_t reserved;
} gtx_disc_event;
#define GTX_DISC_SKIPPED 0u /* no cigar, or in front of the region: the pass does not look at the read */
#define GTX_DISC_COUNTED 1u
#define GTX_DISC_END 2u     /* starts at or behind the region's end: the reference's pass ends here */
typedef struct gtx_disc_read_out
{
  uint32_t first_event, n_events;
  int32_t pos_end; /* region-relative end of the alignment (cov_down) */
  uint32_t state;
} gtx_disc_read_out;
int gtx_disc_create(const char * reference, uint64_t reference_len, int64_t region_begin, int device /* -1: for the host stages only */, gtx_disc ** out);
void gtx_disc_destroy(gtx_disc *);
int gtx_disc_events_batch(gtx_disc *, const uint8_t * d_planes, uint32_t plane_stride, const uint8_t * d_qual, uint32_t qual_stride,
                          const gtx_disc_read * d_reads, const uint32_t * d_cigar, uint32_t n_reads, gtx_disc_event * d_events, uint32_t event_cap,
                          uint32_t * d_counts, gtx_disc_read_out * d_read_out, void * stream);
int gtx_disc_first_pass(const gtx_disc *, const gtx_disc_read * reads, const uint32_t * cigar, const gtx_disc_read_out * read_out, uint32_t n_reads,
                        const gtx_disc_event * events, uint64_t n_events, const uint8_t * seq, uint32_t seq_stride, uint32_t bucket_size,
                        uint32_t * out, uint64_t cap, uint64_t * n_words);

/* The pass to its end (run_first_pass, src/typer/caller.cpp:1186-1365): for every event left by the two filters, the later events
 * within two buckets it travels with -- "ever" (in enough of the reads that cover both, by the phase counts and the coverage
 * between them; any shared read when an indel is involved) and "always" (those at most ten positions on): the sample's haplotype
 * map (HaplotypeInfo, :45-52); the SNPs then leave the buckets.  The result of the file as words: n_indels, per indel the event
 * (pos, type, length, characters), its support fields, file_index where its best support was found, its phase entries; then
 * n_events, per event its "ever" and "always" sets (count + events each).
 * gtx_disc_merge: two such results as one -- merge_haplotypes2 (:64-165) and the union of the files' indels
 * (streamlined_discovery, :2853-2903); `into` may be empty; files are merged in their order.  What the reference does next --
 * realignment of reads to the indels (paw::pairwise_alignment, absent from its tree) -- is not built. */
int gtx_disc_first_pass_haplotypes(const gtx_disc *, const gtx_disc_read * reads, const uint32_t * cigar, const gtx_disc_read_out * read_out,
                                   uint32_t n_reads, const gtx_disc_event * events, uint64_t n_events, const uint8_t * seq, uint32_t seq_stride,
                                   uint32_t bucket_size, int32_t file_index, uint32_t * out, uint64_t cap, uint64_t * n_words);
int gtx_disc_merge(const uint32_t * into, uint64_t n_into, const uint32_t * from, uint64_t n_from, uint32_t * out, uint64_t cap, uint64_t * n_words);

#ifdef __cplusplus
}
#endif
#endif /* GTX_H */
