// kmerust.cpp -- the `kmerust` command line over the HIP path.
// Mirrors src/cli.rs:33-144 (flags, defaults, parse_k messages) and src/main.rs:34-299 (banner on
// stderr unless --quiet, exit 1 on a missing file, warnings for -Q with FASTA / stdin, --save,
// `query`).  Colour escapes are not reproduced (cosmetic).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "kmerust_host.h"

#include <chrono>
#include <unistd.h>

namespace kmerust {

static const char *USAGE =
    "Usage: kmerust [OPTIONS] <K> [PATH]\n"
    "       kmerust query <INDEX> <KMER>\n"
    "\n"
    "Arguments:\n"
    "  <K>     K-mer length (1-32)\n"
    "  [PATH]  Path to a FASTA/FASTQ file (use '-' or omit for stdin) [default: -]\n"
    "\n"
    "Options:\n"
    "  -f, --format <FORMAT>              Output format [default: fasta] [possible values: fasta, tsv, json, histogram]\n"
    "  -m, --min-count <MIN_COUNT>        Minimum count threshold [default: 1]\n"
    "  -q, --quiet                        Suppress informational output\n"
    "  -i, --input-format <INPUT_FORMAT>  Input file format [default: auto] [possible values: auto, fasta, fastq]\n"
    "      --save <SAVE>                  Save k-mer counts to an index file (.kmix, .kmix.gz)\n"
    "  -Q, --min-quality <MIN_QUALITY>    Minimum Phred quality score (0-93) for FASTQ bases\n"
    "      --gpus <N>                     Count on the first N GPUs of the node (RCCL merge of the per-GPU tables) [default: 1]\n"
    "      --devices <LIST>               The same with explicit HIP device ordinals, e.g. 0,2,5\n"
    "  -h, --help                         Print help\n"
    "  -V, --version                      Print version\n";

[[noreturn]] static void usage_error(const std::string &msg) {
    fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", msg.c_str());
    exit(2);  // clap's usage-error status
}

// parse_k, src/cli.rs:103-114
static size_t parse_k(const std::string &s) {
    if (s.empty() || s.find_first_not_of("0123456789") != std::string::npos || s.size() > 19)
        usage_error("invalid value '" + s + "' for '<K>': '" + s + "' is not a valid number");
    const unsigned long long k = strtoull(s.c_str(), nullptr, 10);
    if (k == 0) usage_error("invalid value '" + s + "' for '<K>': k-mer length must be at least 1");
    if (k > 32) usage_error("invalid value '" + s + "' for '<K>': k-mer length must be at most 32");
    return (size_t)k;
}

static uint64_t parse_u64(const std::string &s, const char *what, uint64_t max) {
    if (s.empty() || s.find_first_not_of("0123456789") != std::string::npos || s.size() > 19)
        usage_error("invalid value '" + s + "' for '" + what + "': invalid digit found in string");
    const unsigned long long v = strtoull(s.c_str(), nullptr, 10);
    if (v > max) usage_error("invalid value '" + s + "' for '" + what + "': number too large to fit in target type");
    return v;
}

static int run_query(int argc, char **argv) {
    if (argc != 4) usage_error("the following required arguments were not provided: <INDEX> <KMER>");
    PackedCounts idx;
    try {
        idx = load_index(argv[2]);
    } catch (const Error &e) {
        fprintf(stderr, "Failed to load index:\n %s\n", e.what());
        return 1;
    }
    std::string q = argv[3];
    for (char &c : q) c = (char)toupper((unsigned char)c);
    if (q.size() != idx.k) {
        fprintf(stderr, "Query error:\n k-mer length mismatch: query has %zu bases, index has k=%u\n", q.size(), idx.k);
        return 1;
    }
    uint64_t packed = 0, canon = 0;
    uint32_t pos = 0;
    if (kh_pack(reinterpret_cast<const uint8_t *>(q.data()), idx.k, &packed, &pos) != KH_OK) {
        const unsigned char b = (unsigned char)q[pos];
        if (b >= 0x20 && b < 0x7F) fprintf(stderr, "Invalid k-mer:\n invalid base '%c' (0x%02x) at position %u\n", b, b, pos);
        else fprintf(stderr, "Invalid k-mer:\n invalid base 0x%02x at position %u\n", b, pos);
        return 1;
    }
    kh_canonical(packed, idx.k, &canon, nullptr);
    uint64_t count = 0;
    for (size_t i = 0; i < idx.keys.size(); ++i)
        if (idx.keys[i] == canon) {
            count = idx.counts[i];
            break;
        }
    printf("%llu\n", (unsigned long long)count);
    return 0;
}

// hidden test hook (no device needed): dump the reader's flat batches
static int run_parse_dump(int argc, char **argv) {
    if (argc < 3) return 2;
    SequenceFormat f = SequenceFormat::Auto;
    bool qual = false;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "fasta")) f = SequenceFormat::Fasta;
        else if (!strcmp(argv[i], "fastq")) f = SequenceFormat::Fastq;
        else if (!strcmp(argv[i], "--qual")) qual = true;
    }
    try {
        const uint64_t n = read_sequences(argv[2], f, qual, 1u << 20, [&](const Batch &b) {
            printf("BATCH records=%llu bytes=%zu\n", (unsigned long long)b.records, b.bases.size());
            fwrite(b.bases.data(), 1, b.bases.size(), stdout);
            if (!b.qual.empty()) {
                printf("QUAL\n");
                fwrite(b.qual.data(), 1, b.qual.size(), stdout);
            }
        });
        printf("RECORDS %llu\n", (unsigned long long)n);
    } catch (const Error &e) {
        fprintf(stderr, "Application error:\n %s\n", e.what());
        return 1;
    }
    return 0;
}

int cli_main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "query")) return run_query(argc, argv);  // src/main.rs:39-47
    if (argc > 1 && !strcmp(argv[1], "__parse")) return run_parse_dump(argc, argv);

    std::string k_arg, path = "-", save;
    bool have_k = false, have_path = false, quiet = false;
    OutputFormat fmt = OutputFormat::Fasta;
    const char *fmt_name = "fasta";
    SequenceFormat in_fmt = SequenceFormat::Auto;
    uint64_t min_count = 1;
    int min_quality = -1;
    std::vector<int> devices;  // extension: several GPUs (no counterpart in src/cli.rs)

    auto value_of = [&](int &i, const std::string &arg, const char *name) -> std::string {
        const size_t eq = arg.find('=');
        if (arg.rfind("--", 0) == 0 && eq != std::string::npos) return arg.substr(eq + 1);
        if (arg.rfind("--", 0) != 0 && arg.size() > 2) return arg.substr(2);  // -fVALUE
        if (i + 1 >= argc) usage_error(std::string("a value is required for '") + name + "' but none was supplied");
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const std::string key = a.rfind("--", 0) == 0 ? a.substr(0, a.find('=')) : a.substr(0, 2);
        if (a == "-h" || a == "--help") {
            fputs("MI355X-native canonical k-mer counter (krust-compatible command line)\n\n", stdout);
            fputs(USAGE, stdout);
            return 0;
        } else if (a == "-V" || a == "--version") {
            puts("kmerust 0.3.1 (kmerhip, MI355X)");
            return 0;
        } else if (a == "-q" || a == "--quiet") {
            quiet = true;
        } else if (key == "-f" || key == "--format") {
            const std::string v = value_of(i, a, "--format <FORMAT>");
            if (v == "fasta") fmt = OutputFormat::Fasta, fmt_name = "fasta";
            else if (v == "tsv") fmt = OutputFormat::Tsv, fmt_name = "tsv";
            else if (v == "json") fmt = OutputFormat::Json, fmt_name = "json";
            else if (v == "histogram") fmt = OutputFormat::Histogram, fmt_name = "histogram";
            else usage_error("invalid value '" + v + "' for '--format <FORMAT>'\n  [possible values: fasta, tsv, json, histogram]");
        } else if (key == "-i" || key == "--input-format") {
            const std::string v = value_of(i, a, "--input-format <INPUT_FORMAT>");
            if (v == "auto") in_fmt = SequenceFormat::Auto;
            else if (v == "fasta") in_fmt = SequenceFormat::Fasta;
            else if (v == "fastq") in_fmt = SequenceFormat::Fastq;
            else usage_error("invalid value '" + v + "' for '--input-format <INPUT_FORMAT>'\n  [possible values: auto, fasta, fastq]");
        } else if (key == "-m" || key == "--min-count") {
            min_count = parse_u64(value_of(i, a, "--min-count <MIN_COUNT>"), "--min-count <MIN_COUNT>", UINT64_MAX);
        } else if (key == "-Q" || key == "--min-quality") {
            min_quality = (int)parse_u64(value_of(i, a, "--min-quality <MIN_QUALITY>"), "--min-quality <MIN_QUALITY>", 255);
        } else if (key == "--save") {
            save = value_of(i, a, "--save <SAVE>");
        } else if (key == "--gpus") {
            const uint64_t n = parse_u64(value_of(i, a, "--gpus <N>"), "--gpus <N>", 64);
            if (n == 0) usage_error("invalid value '0' for '--gpus <N>': at least one GPU is needed");
            devices.clear();
            for (uint64_t d = 0; d < n; ++d) devices.push_back((int)d);
        } else if (key == "--devices") {
            const std::string v = value_of(i, a, "--devices <LIST>");
            devices.clear();
            size_t pos = 0;
            while (pos <= v.size()) {
                const size_t comma = std::min(v.find(',', pos), v.size());
                devices.push_back((int)parse_u64(v.substr(pos, comma - pos), "--devices <LIST>", 1023));
                pos = comma + 1;
            }
        } else if (a.size() > 1 && a[0] == '-' && a != "-") {
            usage_error("unexpected argument '" + a + "' found");
        } else if (!have_k) {
            k_arg = a;
            have_k = true;
        } else if (!have_path) {
            path = a;
            have_path = true;
        } else {
            usage_error("unexpected argument '" + a + "' found");
        }
    }
    if (!have_k) usage_error("the following required arguments were not provided:\n  <K>\n\nUsage: kmerust <K> [PATH]");
    const size_t k = parse_k(k_arg);

    const bool from_stdin = is_stdin_path(path);
    if (!from_stdin) {  // src/main.rs:58-67
        struct stat st;
        if (stat(path.c_str(), &st) != 0) {
            fprintf(stderr, "Problem with arguments:\n File not found: %s\n", path.c_str());
            return 1;
        }
    }
    const SequenceFormat resolved = resolve_format(in_fmt, from_stdin ? nullptr : &path);
    if (!quiet) {  // src/main.rs:75-134
        fprintf(stderr, "k-length: %zu\n", k);
        fprintf(stderr, "data: %s\n", from_stdin ? "<stdin>" : path.c_str());
        if (in_fmt == SequenceFormat::Auto) fprintf(stderr, "input-format: %s (auto-detected)\n", format_name(resolved));
        else fprintf(stderr, "input-format: %s\n", format_name(in_fmt));
        fprintf(stderr, "reader: kmerhip\n");
        fprintf(stderr, "output-format: %s\n", fmt_name);
        if (min_count > 1) fprintf(stderr, "min-count: %llu\n", (unsigned long long)min_count);
        if (min_quality >= 0) fprintf(stderr, "min-quality: %d\n", min_quality);
        if (!save.empty()) fprintf(stderr, "save-index: %s\n", save.c_str());
        fprintf(stderr, "\n");
    }
    if (min_quality >= 0 && resolved == SequenceFormat::Fasta)  // src/main.rs:137-143
        fprintf(stderr, "warning: --min-quality is ignored for FASTA input\n");
    if (min_quality >= 0 && from_stdin)  // src/main.rs:145-152
        fprintf(stderr, "warning: --min-quality is not yet supported for stdin input\n");

    const auto t_begin = std::chrono::steady_clock::now();
    struct TimingPrinter {  // KMERUST_TIMING=1: one JSON line on stderr when the command is done (bench.py's `cli` leg reads it)
        std::chrono::steady_clock::time_point t0;
        ~TimingPrinter() {
            const char *e = getenv("KMERUST_TIMING");
            if (!e || !e[0] || e[0] == '0') return;
            const Timing &t = timing();
            const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "{\"kmerust_timing\": {\"total_s\": %.6f, \"create_s\": %.6f, \"read_s\": %.6f, \"push_s\": %.6f, \"finish_s\": %.6f, "
                            "\"result_s\": %.6f, \"write_s\": %.6f, \"buffers_s\": %.6f, \"destroy_s\": %.6f, \"bytes_read\": %llu, \"chunks\": %llu, \"device_record_scan\": %s}}\n",
                    total, t.create_s, t.read_s, t.push_s, t.finish_s, t.result_s, t.write_s, t.buffers_s, t.destroy_s, (unsigned long long)t.bytes_read,
                    (unsigned long long)t.chunks, t.text_path ? "true" : "false");
        }
    } timing_printer{t_begin};
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define KMERUST_UNDER_ASAN 1
#endif
#endif
#if !defined(__SANITIZE_ADDRESS__) && !defined(KMERUST_UNDER_ASAN) && !defined(KMERUST_ALWAYS_CLEAN_EXIT)  // (gcc / clang / make asan)
    leak_at_exit() = !getenv("KMERUST_CLEAN_EXIT");
#endif
    try {
        KmerCounter kc;
        kc.k(k).min_count(min_count).format(fmt).input_format(in_fmt).min_quality(min_quality).devices(devices);
        if (const char *h = getenv("KMERHIP_CAPACITY_HINT")) kc.capacity_hint(strtoull(h, nullptr, 10));
        if (!save.empty()) {  // src/main.rs:155-212: the index holds ALL k-mers, stdout honours --min-count
            const PackedCounts all = kc.count_packed(path, false);
            try {
                save_index(all, save);
            } catch (const Error &e) {
                fprintf(stderr, "Failed to save index:\n %s\n", e.what());
                return 1;
            }
            if (!quiet) fprintf(stderr, "saved: %s (%zu k-mers)\n", save.c_str(), all.keys.size());
            write_counts(stdout, all, fmt, min_count);
        } else {
            kc.run(path);
        }
    } catch (const Error &e) {
        fprintf(stderr, "Application error:\n %s\n", e.what());
        return 1;
    }
    return 0;
}

}  // namespace kmerust

int main(int argc, char **argv) {
    const int rc = kmerust::cli_main(argc, argv);
    if (kmerust::leak_at_exit()) {  // one count per process: leave without the runtime's piecemeal teardown (see kmerust_host.cpp)
        fflush(nullptr);
        _exit(rc);
    }
    return rc;
}
