// tdc_main.cpp -- mini `tdc` command line with the reference driver's contract for this path
// (src/tudocomp_driver/tudocomp_driver.cpp:52-395, include/tudocomp_driver/Options.hpp:15-32):
//   tdc -a 'lcpcomp(coder=huff,threshold=2)' [-o OUT] [-f] [--raw] [--stats] FILE     compress (on the GPU)
//   tdc -d [-a ALGO] [--raw] [-o OUT] [-f] FILE                                       decompress (host)
//   tdc -l                                                                            list registered algorithms
// Compressed files start with "<algorithm id>%" unless --raw; default output is FILE.tdc; exit codes 0 / 1.
#include "tdc_amd.hpp"

#include <getopt.h>
#include <sys/stat.h>
#include <chrono>
#include <cstdio>
#include <iostream>

using namespace tdc_amd;

static bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
[[noreturn]] static void fail(const std::string& msg) { std::cerr << "Error: " << msg << std::endl; exit(1); }

int main(int argc, char** argv) {
    static const option OPTS[] = {
        {"algorithm", required_argument, nullptr, 'a'}, {"decompress", no_argument, nullptr, 'd'},
        {"force", no_argument, nullptr, 'f'},           {"list", no_argument, nullptr, 'l'},
        {"output", required_argument, nullptr, 'o'},    {"stats", no_argument, nullptr, 's'},
        {"raw", no_argument, nullptr, 1001},            {"help", no_argument, nullptr, 1000},
        {"device", required_argument, nullptr, 1002},   {0, 0, 0, 0}};
    std::string algo, ofile;
    bool decompress = false, force = false, list = false, stats = false, raw = false;
    int device = 0;
    for (int c; (c = getopt_long(argc, argv, "a:dflo:s", OPTS, nullptr)) != -1;) {
        switch (c) {
            case 'a': algo = optarg; break;
            case 'd': decompress = true; break;
            case 'f': force = true; break;
            case 'l': list = true; break;
            case 'o': ofile = optarg; break;
            case 's': stats = true; break;
            case 1001: raw = true; break;
            case 1002: device = atoi(optarg); break;
            case 1000:
                std::cout << "Usage: tdc [-a ALGORITHM] [-d] [-f] [-o OUTPUT] [--raw] [--stats] [--device N] FILE\n       tdc -l\n";
                return 0;
            default: return 1;
        }
    }
    try {
        if (list) {
            std::cout << "This build of tdc contains the following compressors:\n";
            for (auto& s : registered_algorithms()) std::cout << "  " << s << "\n";
            return 0;
        }
        if (optind >= argc) fail("No input file given (see --help)");
        const std::string file = argv[optind];
        if (!decompress && algo.empty()) fail("No algorithm given (-a)");
        if (ofile.empty()) ofile = decompress ? file + ".out" : file + ".tdc";           // tudocomp_driver.cpp:30,163
        if (file_exists(ofile) && !force) fail("Output file " + ofile + " already exists (use -f to overwrite)");

        Input inp = Input::from_file(file);
        const size_t in_size = inp.size();
        bytes result;
        Output out(result);
        const auto t0 = std::chrono::steady_clock::now();
        Selection sel;
        if (!decompress) {
            sel = select_algorithm(algo, nullptr, device);
            if (!raw) {                                                                  // :261-266
                if (algo.find('%') != std::string::npos) fail("algorithm id must not contain '%'");
                result.insert(result.end(), algo.begin(), algo.end());
                result.push_back('%');
            }
            if (sel.restrictions.has_restrictions()) inp = Input(inp, sel.restrictions);  // :268-270
            sel.compressor->compress(inp, out);
        } else {
            std::string header;
            if (!raw) {                                                                  // :284-313
                const bytes& d = inp.raw();
                size_t i = 0; bool found = false;
                for (; i < d.size() && i <= 1023; ++i) { if (d[i] == '%') { found = true; break; } header.push_back((char)d[i]); }
                if (!found) fail("Input did not have an algorithm header!");
                inp = Input(inp, header.size() + 1);
            }
            const std::string id = !algo.empty() ? algo : header;
            if (id.empty()) fail("No algorithm given (-a) and no header present");
            sel = select_algorithm(id);
            if (sel.restrictions.has_restrictions()) out = Output(out, sel.restrictions);  // :336-338
            sel.compressor->decompress(inp, out);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::ofstream of(ofile, std::ios::binary | std::ios::trunc);
        if (!of) fail("Could not open " + ofile + " for writing");
        of.write((const char*)result.data(), (std::streamsize)result.size());
        if (stats) {                                                                      // :361-391 (same meta keys)
            std::printf("{\"meta\":{\"config\":\"%s\",\"input\":\"%s\",\"inputSize\":%zu,\"output\":\"%s\",\"outputSize\":%zu,"
                        "\"rate\":%.6f},\"timeTotalMs\":%.3f", algo.c_str(), file.c_str(), in_size, ofile.c_str(), result.size(),
                        in_size ? (double)result.size() / (double)in_size : 0.0, ms);
            if (auto* c = decompress ? nullptr : dynamic_cast<LCPCompressor*>(sel.compressor.get())) {
                const tdc_gpu_stats& st = c->last_stats;
                std::printf(",\"stats\":{\"factors\":%llu,\"maxlcp\":%llu,\"entries\":%llu,\"num_flattened\":%llu,\"max_depth_lb\":%llu,"
                            "\"gpu_ms\":{\"sa\":%.3f,\"phi\":%.3f,\"plcp\":%.3f,\"factorize\":%.3f,\"flatten\":%.3f,\"encode\":%.3f,\"total\":%.3f}}",
                            (unsigned long long)st.factors, (unsigned long long)st.maxlcp, (unsigned long long)st.entries,
                            (unsigned long long)st.num_flattened, (unsigned long long)st.max_depth_lb, st.ms_sa, st.ms_phi, st.ms_plcp,
                            st.ms_factorize, st.ms_flatten, st.ms_encode, st.ms_total);
            }
            std::printf("}\n");
        }
        return 0;
    } catch (const std::exception& e) {
        fail(e.what());                                                                   // :392-395
    }
}
