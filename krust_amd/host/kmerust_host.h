// kmerust_host.h -- C++ host side above the C ABI (include/kmerhip.h).
//
// The reference's toolchain (Rust) is absent from this image, so the host layer that would be the
// `kmerust` crate is written in C++ and mirrors the reference's interface for this path:
//   SequenceFormat / from_extension / resolve      src/format.rs:47-102
//   Input ("-" = stdin)                            src/input.rs:28-70
//   reader::read / read_with_quality               src/reader.rs:58-79,82-144,167-247 (FASTA/FASTQ, gzip)
//   KmerCounter builder                            src/builder.rs:95-526
//   output_counts (fasta / tsv / json / histogram) src/run.rs:441-486
//   KMIX index save / load / query                 src/index.rs:7-23,222-431
//   CLI                                            src/cli.rs:33-144, src/main.rs:34-299
// Counting itself always goes through kh_* (the HIP path); there is no CPU counting here.
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/kmerhip.h"

namespace kmerust {

enum class OutputFormat { Fasta, Tsv, Json, Histogram };  // src/cli.rs:90-101
enum class SequenceFormat { Auto, Fasta, Fastq };         // src/format.rs:20-33

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct KmerLengthError : Error {  // src/error.rs:86-95
    explicit KmerLengthError(size_t k)
        : Error("k-mer length " + std::to_string(k) + " is out of range: must be between 1 and 32"), k(k) {}
    size_t k;
};

// ---- format / input -------------------------------------------------------------------------
SequenceFormat format_from_extension(const std::string &path);          // src/format.rs:47-70
SequenceFormat resolve_format(SequenceFormat f, const std::string *path);  // src/format.rs:97-102
const char *format_name(SequenceFormat f);                                 // Display, src/format.rs:117-125
inline bool is_stdin_path(const std::string &p) { return p == "-"; }      // src/input.rs:55-61

// ---- reader ----------------------------------------------------------------------------------
// Streams a FASTA/FASTQ file (plain or gzip; "-" = stdin) into flat batches: records separated by
// '\n' in `bases`, and, when want_qual, a parallel `qual` buffer.  `sink` is called with whole
// records only (k-mers never span records).  Returns the number of records.
// Parsing choices where the reference's parser (rust-bio 3.0.0) is not pinned by its tests are
// listed in DESIGN.md: lines are right-trimmed of CR / blanks, FASTA sequence lines are
// concatenated, FASTQ records may wrap over several lines, a record header must start with '>' /
// '@', and a FASTQ quality string must be as long as its sequence.
struct Batch {
    std::vector<uint8_t> bases, qual;
    uint64_t records = 0;
};
using BatchSink = std::function<void(const Batch &)>;
uint64_t read_sequences(const std::string &path, SequenceFormat fmt, bool want_qual, size_t batch_bytes,
                        const BatchSink &sink);

// ---- counting --------------------------------------------------------------------------------
struct PackedCounts {
    uint32_t k = 0;
    std::vector<uint64_t> keys, counts;  // packed canonical key, count
};

// The fluent builder of src/builder.rs:95-526 (same option names and defaults).
class KmerCounter {
public:
    KmerCounter &k(size_t k);  // throws KmerLengthError outside 1..=32 (builder.rs:120-128)
    KmerCounter &min_count(uint64_t n) { min_count_ = n; return *this; }
    KmerCounter &format(OutputFormat f) { format_ = f; return *this; }
    KmerCounter &input_format(SequenceFormat f) { input_format_ = f; return *this; }
    KmerCounter &min_quality(int q) { min_quality_ = q; return *this; }  // -1 = None
    KmerCounter &capacity_hint(uint64_t n) { capacity_hint_ = n; return *this; }
    KmerCounter &device(int d) { device_ = d; return *this; }
    // No reference counterpart (the reference is one process on CPU cores): count on several GPUs of the node,
    // one table per device, merged by the library's RCCL exchange (kh_group_*) into a table sharded by hash range.
    KmerCounter &devices(std::vector<int> d) { devices_ = std::move(d); return *this; }

    // count(): HashMap<String,u64> filtered by min_count (builder.rs:242-262)
    std::unordered_map<std::string, uint64_t> count(const std::string &path) const;
    // packed keys, UNfiltered unless apply_min_count (count_kmers_from_sequences shape, streaming.rs:198-204)
    PackedCounts count_packed(const std::string &path, bool apply_min_count = false) const;
    // histogram(): count -> frequency, ascending (builder.rs:289-307, histogram.rs:88-94)
    std::vector<std::pair<uint64_t, uint64_t>> histogram(const std::string &path) const;
    // run(): count and write to stdout in the configured format (builder.rs:366-373)
    void run(const std::string &path) const;
    // count_to_writer() (builder.rs:403-460)
    void count_to_writer(const std::string &path, FILE *out) const;

    size_t get_k() const { return k_; }

private:
    friend struct Session;
    size_t k_ = 0;
    bool k_set_ = false;
    uint64_t min_count_ = 1;
    OutputFormat format_ = OutputFormat::Fasta;
    SequenceFormat input_format_ = SequenceFormat::Auto;
    int min_quality_ = -1;
    uint64_t capacity_hint_ = 0;
    int device_ = -1;
    std::vector<int> devices_;
};

// ---- phase walls of the last count (KMERUST_TIMING=1 makes the CLI print them as one JSON line on stderr) ---------
// Stands where the reference has its tracing spans "read_sequences" / "process_sequences" / "unpack_kmers"
// (src/run.rs:253-279, feature `tracing`).
struct Timing {
    double create_s = 0;   // device context(s): runtime start-up, table allocation
    double read_s = 0;     // file -> host buffer (pread / gzread / line parser)
    double push_s = 0;     // kh_push_text / kh_push: H2D, record scan, counting of the chunk
    double finish_s = 0;   // kh_finish (+ the multi-GPU merge)
    double result_s = 0;   // kh_histogram / kh_result_copy
    double write_s = 0;    // formatting and writing the output
    double buffers_s = 0;  // pinned chunk buffers (kh_host_alloc)
    double destroy_s = 0;  // releasing the device context(s)
    uint64_t bytes_read = 0, chunks = 0;
    bool text_path = false;  // records were found on the device (kh_push_text)
};
Timing &timing();
bool &leak_at_exit();  // the CLI sets it: device contexts are not torn down before the process ends

// ---- output (src/run.rs:441-486) -------------------------------------------------------------
std::string unpack_to_string(uint64_t bits, uint32_t k);  // src/kmer.rs:451-456
void write_counts(FILE *out, const PackedCounts &pc, OutputFormat fmt, uint64_t min_count);
void write_histogram(FILE *out, const std::vector<std::pair<uint64_t, uint64_t>> &hist);

// ---- KMIX index (src/index.rs) -----------------------------------------------------------------
uint32_t crc32_ieee(const uint8_t *data, size_t n, uint32_t crc = 0);  // src/index.rs:404-431
void save_index(const PackedCounts &pc, const std::string &path);      // gzip if path ends in .gz
PackedCounts load_index(const std::string &path);                      // validates magic/version/k/size/CRC

// ---- CLI ---------------------------------------------------------------------------------------
int cli_main(int argc, char **argv);

}  // namespace kmerust
