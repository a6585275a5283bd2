// nohuman_cli.cpp -- C++ host with the observable behaviour of the reference's `nohuman` binary
// (the reference host is Rust, which cannot be built in this image): same
// flags, log lines, output naming and temp-dir contract, with the kraken2 subprocess replaced by
// nh_run() from libnohuman_engine.so.
//
// Mirrors, file:line under /root/reference:
//   flags                      src/main.rs:21-104         (clap Args)
//   logger format / level      src/main.rs:110-121        ([<UTC>Z LEVEL ] message on stderr)
//   early exits                src/main.rs:123-195        (--list-db-versions, --download, --check)
//   database resolution        src/main.rs:388-434, src/download.rs:178-232, src/lib.rs:119-141
//   output codec decision      src/main.rs:238-245, src/compression.rs:107-118,271-296
//   temp dir + kraken_out[#]   src/main.rs:248-265
//   default output names       src/main.rs:273-338        (incl. the `.zstd` extension quirk)
//   compress stage             src/main.rs:342-368, src/compression.rs:182-268
//   summary line               src/lib.rs:38-45
// Not rebuilt (network provisioning, out of scope): --download / --list-db-versions report so.
#include <dirent.h>
#include <errno.h>
#include <pwd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <string>
#include <thread>
#include <vector>

#include "nohuman_engine.h"

static bool g_verbose = false;

static void logmsg(const char *level, const char *fmt, ...) {
    if (!g_verbose && strcmp(level, "DEBUG") == 0) return;
    char ts[32];
    time_t t = time(nullptr);
    struct tm tmv;
    gmtime_r(&t, &tmv);
    strftime(ts, sizeof ts, "%Y-%m-%dT%H:%M:%SZ", &tmv);
    fprintf(stderr, "[%s %-5s] ", ts, level);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
#define INFO(...) logmsg("INFO", __VA_ARGS__)
#define DEBUG(...) logmsg("DEBUG", __VA_ARGS__)
#define WARN(...) logmsg("WARN", __VA_ARGS__)
#define ERROR(...) logmsg("ERROR", __VA_ARGS__)

[[noreturn]] static void die(const char *fmt, ...) {  // anyhow's "Error: ..." on stderr, exit 1
    fprintf(stderr, "Error: ");
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    exit(1);
}

// ---- path helpers with std::path semantics -----------------------------------------------------------
static bool exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}
static bool is_dir(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
static std::string parent_of(const std::string &p) {
    size_t s = p.find_last_of('/');
    if (s == std::string::npos) return "";
    return s == 0 ? "/" : p.substr(0, s);
}
static std::string file_name(const std::string &p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}
static std::string join(const std::string &dir, const std::string &name) {
    if (dir.empty()) return name;
    return dir.back() == '/' ? dir + name : dir + "/" + name;
}
// Path::extension: text after the last '.' of the file name, none for ".hidden" or no dot
static std::string extension(const std::string &p) {
    std::string f = file_name(p);
    size_t d = f.find_last_of('.');
    if (d == std::string::npos || d == 0) return "";
    return f.substr(d + 1);
}
static std::string file_stem(const std::string &p) {
    std::string f = file_name(p);
    size_t d = f.find_last_of('.');
    if (d == std::string::npos || d == 0) return f;
    return f.substr(0, d);
}
static std::string quoted(const std::string &p) { return "\"" + p + "\""; }  // {:?} of a path

// ---- CompressionFormat (src/compression.rs) ----------------------------------------------------------
enum Codec { C_NONE, C_BZ2, C_GZ, C_XZ, C_ZST };
static const char *codec_ext(Codec c) {
    switch (c) {
        case C_BZ2: return "bz2";
        case C_GZ: return "gz";
        case C_XZ: return "xz";
        case C_ZST: return "zst";
        default: return "";
    }
}
static bool codec_from_str(const std::string &s, Codec &c) {  // FromStr, case-insensitive single letter
    if (s.size() != 1) return false;
    switch (s[0]) {
        case 'b': case 'B': c = C_BZ2; return true;
        case 'g': case 'G': c = C_GZ; return true;
        case 'x': case 'X': c = C_XZ; return true;
        case 'z': case 'Z': c = C_ZST; return true;
        case 'u': case 'U': c = C_NONE; return true;
        default: return false;
    }
}
static Codec codec_from_path(const std::string &p) {
    std::string e = extension(p);
    if (e == "bz2") return C_BZ2;
    if (e == "gz") return C_GZ;
    if (e == "xz") return C_XZ;
    if (e == "zst" || e == "zstd") return C_ZST;
    return C_NONE;
}
static Codec codec_from_magic(const std::string &p) {
    unsigned char m[5] = {0, 0, 0, 0, 0};
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) die("Failed to open %s", quoted(p).c_str());
    size_t got = fread(m, 1, 5, f);
    fclose(f);
    if (got != 5) die("Failed to read the first five bytes of the file");
    if (m[0] == 0x1f && m[1] == 0x8b) return C_GZ;
    if (m[0] == 0x42 && m[1] == 0x5a) return C_BZ2;
    if (m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) return C_ZST;
    if (m[0] == 0xfd && m[1] == 0x37 && m[2] == 0x7a && m[3] == 0x58 && m[4] == 0x5a) return C_XZ;
    return C_NONE;
}
static std::string add_extension(Codec c, const std::string &p) {
    if (c == C_NONE) return p;
    return p + "." + codec_ext(c);  // "a.fq" -> "a.fq.gz", "a" -> "a.gz"
}

// compress stage (src/compression.rs:182-268): content parity, not byte-identical streams
// ---- database discovery (src/download.rs:178-232, src/lib.rs:119-141) --------------------------------
static bool validate_db_directory(const std::string &p, std::string &actual) {
    const char *req[3] = {"hash.k2d", "opts.k2d", "taxo.k2d"};
    for (const std::string &d : {p, join(p, "db")}) {
        bool ok = is_dir(d);
        for (const char *r : req) ok = ok && exists(join(d, r));
        if (ok) {
            actual = d;
            return true;
        }
    }
    return false;
}
struct Installed {
    std::string version, path, added;
};
static bool read_metadata(const std::string &dir, Installed &m) {  // nohuman-db.toml: version, added
    FILE *f = fopen(join(dir, "nohuman-db.toml").c_str(), "r");
    if (!f) return false;
    char line[1024];
    bool v = false, a = false;
    while (fgets(line, sizeof line, f)) {
        char key[64], val[512];
        if (sscanf(line, " %63[A-Za-z_] = \"%511[^\"]\"", key, val) == 2) {
            if (!strcmp(key, "version")) { m.version = val; v = true; }
            if (!strcmp(key, "added")) { m.added = val; a = true; }
        }
    }
    fclose(f);
    return v && a;
}
static std::string date_key(const std::string &s) {  // invalid dates sort as the legacy date
    int y, mo, d;
    if (s.size() == 10 && sscanf(s.c_str(), "%4d-%2d-%2d", &y, &mo, &d) == 3) return s;
    return "1970-01-01";
}
static std::vector<Installed> installed_databases(const std::string &root) {
    std::vector<Installed> out;
    if (DIR *d = opendir(root.c_str())) {
        while (struct dirent *e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            std::string p = join(root, e->d_name), actual;
            Installed m;
            if (!is_dir(p) || !read_metadata(p, m)) continue;
            if (validate_db_directory(p, actual)) {
                m.path = p;
                out.push_back(m);
            } else {
                DEBUG("Skipping %s because the required Kraken files were not found", quoted(p).c_str());
            }
        }
        closedir(d);
    }
    std::string actual;
    Installed tmp;
    if (validate_db_directory(root, actual) && !read_metadata(root, tmp))
        out.push_back({"legacy", root, "1970-01-01"});
    return out;
}

struct Args {
    std::vector<std::string> input;
    std::string out1, out2, database, db_version, kraken_output, kraken_report;
    bool has_out1 = false, has_out2 = false, check = false, download = false, list = false, human = false;
    bool has_type = false;
    Codec type = C_NONE;
    unsigned threads = 1;
    float confidence = 0.0f;
    std::string conf_text = "0.0";
};

static void usage(FILE *f) {
    fputs("Usage: nohuman [OPTIONS] [INPUT]...\n\n"
          "Arguments:\n  [INPUT]...  Input file(s) to remove human reads from\n\n"
          "Options:\n"
          "  -o, --out1 <OUTPUT_1>        First output file\n"
          "  -O, --out2 <OUTPUT_2>        Second output file\n"
          "  -c, --check                  Check that all required dependencies are available and exit\n"
          "  -d, --download               Download the database\n"
          "  -D, --db <PATH>              Path to the database [env: NOHUMAN_DB=] [default: ~/.nohuman/db]\n"
          "      --db-version <VERSION>   Name of the database version to use\n"
          "      --list-db-versions       List available database versions and exit\n"
          "  -F, --output-type <FORMAT>   Output compression format. u: uncompressed; b: Bzip2; g: Gzip; x: Xz (Lzma); z: Zstd\n"
          "  -t, --threads <INT>          Number of threads to use [default: 1]\n"
          "  -H, --human                  Output human reads instead of removing them\n"
          "  -C, --conf <[0, 1]>          Kraken2 minimum confidence score [default: 0.0]\n"
          "  -k, --kraken-output <FILE>   Write the Kraken2 read classification output to a file\n"
          "  -r, --kraken-report <FILE>   Write the Kraken2 report with aggregate counts/clade to file\n"
          "  -v, --verbose                Set the logging level to verbose\n"
          "  -h, --help                   Print help\n"
          "  -V, --version                Print version\n",
          f);
}

[[noreturn]] static void arg_error(const char *fmt, ...) {  // clap: message, exit code 2
    fprintf(stderr, "error: ");
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n\nFor more information, try '--help'.\n");
    exit(2);
}

static Args parse_args(int argc, char **argv) {
    Args a;
    const char *home = getenv("HOME");
    if (!home) {
        struct passwd *pw = getpwuid(getuid());
        home = pw ? pw->pw_dir : "";
    }
    a.database = join(join(home, ".nohuman"), "db");
    if (const char *env = getenv("NOHUMAN_DB")) a.database = env;
    auto value = [&](int &i, const std::string &flag) -> std::string {
        if (i + 1 >= argc) arg_error("a value is required for '%s' but none was supplied", flag.c_str());
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i], v;
        size_t eq = s.find('=');
        bool inline_val = s.rfind("--", 0) == 0 && eq != std::string::npos;
        if (inline_val) {
            v = s.substr(eq + 1);
            s = s.substr(0, eq);
        }
        auto val = [&]() { return inline_val ? v : value(i, s); };
        if (s == "-o" || s == "--out1") { a.out1 = val(); a.has_out1 = true; }
        else if (s == "-O" || s == "--out2") { a.out2 = val(); a.has_out2 = true; }
        else if (s == "-c" || s == "--check") a.check = true;
        else if (s == "-d" || s == "--download") a.download = true;
        else if (s == "-D" || s == "--db") a.database = val();
        else if (s == "--db-version") a.db_version = val();
        else if (s == "--list-db-versions") a.list = true;
        else if (s == "-F" || s == "--output-type") {
            std::string t = val();
            if (!codec_from_str(t, a.type)) arg_error("invalid value '%s' for '--output-type <FORMAT>': %s is not a valid output format", t.c_str(), t.c_str());
            a.has_type = true;
        } else if (s == "-t" || s == "--threads") {
            std::string t = val();
            char *end;
            long n = strtol(t.c_str(), &end, 10);
            if (*end || t.empty() || n <= 0) arg_error("invalid value '%s' for '--threads <INT>': number would be zero for non-zero type", t.c_str());
            a.threads = (unsigned)n;
        } else if (s == "-H" || s == "--human") a.human = true;
        else if (s == "-C" || s == "--conf") {
            std::string t = val();
            char *end;
            float c = strtof(t.c_str(), &end);
            if (*end || t.empty()) arg_error("invalid value '%s' for '--conf <[0, 1]>': Confidence score must be a number", t.c_str());
            if (!(c >= 0.0f && c <= 1.0f)) arg_error("invalid value '%s' for '--conf <[0, 1]>': Confidence score must be in the closed interval [0, 1]", t.c_str());
            a.confidence = c;
        } else if (s == "-k" || s == "--kraken-output") a.kraken_output = val();
        else if (s == "-r" || s == "--kraken-report") a.kraken_report = val();
        else if (s == "-v" || s == "--verbose") g_verbose = true;
        else if (s == "-h" || s == "--help") { usage(stdout); exit(0); }
        else if (s == "-V" || s == "--version") { puts("nohuman 0.5.1 (MI355X engine)"); exit(0); }
        else if (s.size() > 1 && s[0] == '-') arg_error("unexpected argument '%s' found", s.c_str());
        else {
            if (!exists(s)) arg_error("invalid value '%s' for '[INPUT]...': %s does not exist", s.c_str(), quoted(s).c_str());
            a.input.push_back(s);
        }
    }
    if (a.input.empty() && !a.check && !a.download && !a.list)
        arg_error("the following required arguments were not provided:\n  <INPUT>...");
    return a;
}

// f32 Display as Rust prints it (shortest round-trip decimal), then what kraken2 would parse (f64)
static double confidence_as_kraken2_sees_it(float c, std::string &text) {
    char buf[64];
    for (int prec = 1; prec < 12; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, (double)c);
        if (strtof(buf, nullptr) == c) break;
    }
    text = buf;
    return strtod(buf, nullptr);
}

static std::string default_out_name(const std::string &in, Codec out_codec) {
    // src/main.rs:274-290: strip a trailing compression extension only when it equals the codec's
    // canonical extension (so ".zstd" is not stripped), then "<stem>.nohuman.fq[.ext]" beside the input
    std::string ext = codec_ext(codec_from_path(in));
    std::string stem;
    if (extension(in) == ext) {
        std::string no_ext = in.substr(0, in.size() - (extension(in).empty() ? 0 : extension(in).size() + 1));
        stem = file_stem(no_ext);
    } else {
        stem = file_stem(in);
    }
    return add_extension(out_codec, join(parent_of(in), stem + ".nohuman.fq"));
}

int main(int argc, char **argv) {
    Args args = parse_args(argc, argv);
    if (args.list) die("Failed to download database manifest: network access is not available in this build");
    if (args.download) {
        INFO("Downloading database...");
        die("Failed to download database: network access is not available in this build");
    }
    // dependency check: the engine library + a gfx950 device take the place of `kraken2` on PATH
    char probe[256];
    if (nh_probe(probe, sizeof probe) != 0) {
        DEBUG("kraken2 is not executable");
        ERROR("The following dependencies are missing:");
        ERROR("kraken2 (in-process engine: %s)", probe);
        die("Missing dependencies");
    }
    DEBUG("kraken2 is executable (in-process engine: %s)", probe);
    if (args.check) {
        INFO("All dependencies are available");
        return 0;
    }
    if (args.input.empty()) die("No input files provided");
    if (args.input.size() > 2) die("Only one or two input files are allowed");

    // resolve_database (src/main.rs:393-434)
    std::string db_path, db_ver;
    if (!args.db_version.empty()) {
        if (args.db_version == "all")
            die("Cannot run with `--db-version all`. Use `--download --db-version all` to download every database.");
        bool found = false;
        for (const Installed &i : installed_databases(args.database))
            if (i.version == args.db_version) {
                if (!validate_db_directory(i.path, db_path))
                    die("Required files (hash.k2d, opts.k2d, taxo.k2d) not found in %s or its 'db' subdirectory", quoted(i.path).c_str());
                db_ver = i.version;
                found = true;
                break;
            }
        if (!found)
            die("Database version '%s' is not installed under %s. Run `nohuman --download --db-version %s` to download it.",
                args.db_version.c_str(), quoted(args.database).c_str(), args.db_version.c_str());
    } else if (!validate_db_directory(args.database, db_path)) {
        std::vector<Installed> inst = installed_databases(args.database);
        const Installed *best = nullptr;
        for (const Installed &i : inst)
            if (!best || date_key(i.added) >= date_key(best->added)) best = &i;
        if (!best)
            die("Database does not exist at %s. Run `nohuman --download` to fetch one.", quoted(args.database).c_str());
        if (!validate_db_directory(best->path, db_path))
            die("Required files (hash.k2d, opts.k2d, taxo.k2d) not found in %s or its 'db' subdirectory", quoted(best->path).c_str());
        db_ver = best->version;
    }
    if (!db_ver.empty())
        INFO("Using database version %s at %s", db_ver.c_str(), quoted(db_path).c_str());
    else
        INFO("Using database at %s", quoted(db_path).c_str());

    const bool paired = args.input.size() == 2;
    Codec out_codec = args.has_type ? args.type : args.has_out1 ? codec_from_path(args.out1) : codec_from_magic(args.input[0]);

    // The reference lets kraken2 write kraken_out*.fq into a temporary directory "nohuman*" in the current
    // directory (src/main.rs:248-257) and compresses them to the output paths afterwards
    // (src/main.rs:342-368).  Here the engine's writer feeds the kept records straight into the output
    // encoder: no temporary file, one pass.
    std::string out1 = args.has_out1 ? args.out1 : default_out_name(args.input[0], out_codec);
    std::string out2 = paired ? (args.has_out2 ? args.out2 : default_out_name(args.input[1], out_codec)) : "";
    const int codec = out_codec == C_GZ ? NH_CODEC_GZIP : out_codec == C_BZ2 ? NH_CODEC_BZIP2 : out_codec == C_XZ ? NH_CODEC_XZ
                    : out_codec == C_ZST ? NH_CODEC_ZSTD : NH_CODEC_NONE;
    INFO(args.human ? "Keeping human reads..." : "Removing human reads...");

    std::string conf_text;
    const double conf64 = confidence_as_kraken2_sees_it(args.confidence, conf_text);
    DEBUG("Running kraken2...");
    DEBUG("With arguments: [\"--threads\", \"%u\", \"--db\", %s, \"--output\", %s, \"--confidence\", \"%s\"%s%s]",
          args.threads, quoted(db_path).c_str(),
          quoted(args.kraken_output.empty() ? "/dev/null" : args.kraken_output).c_str(), conf_text.c_str(),
          paired ? ", \"--paired\"" : "", args.human ? ", \"--classified-out\", ..." : ", \"--unclassified-out\", ...");
    nh_run_args ra;
    memset(&ra, 0, sizeof ra);
    ra.db_dir = db_path.c_str();
    ra.in1 = args.input[0].c_str();
    ra.in2 = paired ? args.input[1].c_str() : nullptr;
    // The engine writes to "<out>.partial" and the finished file is renamed over the output path: a run
    // that fails -- before or after it created anything -- leaves a file already at the output path
    // untouched, and `-o` equal to an input path works as in the reference (the input is replaced after
    // it has been read: src/main.rs:342-368 compresses the temporary kraken_out*.fq to the output path
    // last).  Outputs that exist and are not regular files (/dev/stdout, a fifo) are written directly.
    auto staged = [](const std::string &out) {
        struct stat sb;
        if (stat(out.c_str(), &sb) == 0 && !S_ISREG(sb.st_mode)) return out;
        return out + ".partial";
    };
    const std::string part1 = staged(out1), part2 = paired ? staged(out2) : "";
    ra.out1 = part1.c_str();
    ra.out2 = paired ? part2.c_str() : nullptr;
    ra.out_codec = codec;
    ra.codec_threads = paired ? (args.threads / 2 ? args.threads / 2 : 1) : args.threads;  // src/main.rs:342-346
    ra.kraken_output = args.kraken_output.empty() ? "/dev/null" : args.kraken_output.c_str();
    ra.report = args.kraken_report.empty() ? nullptr : args.kraken_report.c_str();
    ra.confidence = conf64;
    ra.threads = args.threads;
    ra.keep_human = args.human ? 1 : 0;
    ra.n_devices = 0;  // all visible devices, database replicated (env NOHUMAN_DEVICES narrows it)
    std::vector<int32_t> devs;
    if (const char *dv = getenv("NOHUMAN_DEVICES")) {
        for (const char *p = dv; *p;) {
            devs.push_back((int32_t)strtol(p, (char **)&p, 10));
            while (*p == ',' || *p == ' ') p++;
        }
        ra.n_devices = (int32_t)devs.size();
        ra.device_ids = devs.data();
    }
    nh_stats st;
    if (nh_run(&ra, &st) != 0) {
        std::string msg = nh_last_error();
        // nothing half-written stays behind, and nothing this run did not create is touched
        if (part1 != out1) unlink(part1.c_str());
        if (paired && part2 != out2) unlink(part2.c_str());
        die("Failed to run kraken2\n\nCaused by:\n    kraken2 failed with stderr %s", msg.c_str());
    }
    if ((part1 != out1 && rename(part1.c_str(), out1.c_str()) != 0) ||
        (paired && part2 != out2 && rename(part2.c_str(), out2.c_str()) != 0)) {
        const std::string why = strerror(errno);
        if (part1 != out1) unlink(part1.c_str());
        if (paired && part2 != out2) unlink(part2.c_str());
        die("Failed to move the output into place: %s", why.c_str());
    }
    // src/lib.rs:38-45 (0/0 prints NaN there as well)
    auto pct = [&](uint64_t a) -> std::string {
        if (st.total_sequences == 0) return "NaN";
        char b[32];
        snprintf(b, sizeof b, "%.2f", (double)a / (double)st.total_sequences * 100.0);
        return b;
    };
    INFO("%llu / %llu (%s%%) sequences classified as human; %llu (%s%%) as non-human",
         (unsigned long long)st.classified, (unsigned long long)st.total_sequences, pct(st.classified).c_str(),
         (unsigned long long)st.unclassified, pct(st.unclassified).c_str());
    INFO("Kraken2 finished. Organising output...");

    if (paired && args.threads / 2 > 1) {  // the reference announces both files before its two compress threads run
        INFO("Writing output file to: %s", quoted(out1).c_str());
        INFO("Writing output file to: %s", quoted(out2).c_str());
    } else {
        INFO("Output file written to: %s", quoted(out1).c_str());
        if (paired) INFO("Output file written to: %s", quoted(out2).c_str());
    }
    if (!args.kraken_output.empty() && args.kraken_output != "/dev/null")
        INFO("Kraken output file written to: %s", quoted(args.kraken_output).c_str());
    if (!args.kraken_report.empty()) INFO("Kraken report file written to: %s", quoted(args.kraken_report).c_str());
    INFO("Done.");
    return 0;
}
