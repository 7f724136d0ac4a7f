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
        {"device", required_argument, nullptr, 1002},   {"blocks", required_argument, nullptr, 1003},
        {"devices", required_argument, nullptr, 1004},  {0, 0, 0, 0}};
    std::string algo, ofile;
    bool decompress = false, force = false, list = false, stats = false, raw = false;
    int device = 0, ndev = 0;
    size_t block_size = 0;
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
            case 1003: block_size = (size_t)strtoull(optarg, nullptr, 10); break;
            case 1004: ndev = atoi(optarg); break;
            case 1000:
                std::cout << "Usage: tdc [-a ALGORITHM] [-d] [-f] [-o OUTPUT] [--raw] [--stats] [--device N] [--blocks BYTES [--devices N]] FILE\n       tdc -l\n"
                             "  --blocks BYTES  lcpcomp only: cut the input into independent blocks of BYTES bytes, spread them over the visible\n"
                             "                  GPUs (or the first N with --devices) and frame the streams in a block container; -d detects it\n";
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
            if (block_size) {                                                            // block mode: restrictions are applied per block
                auto* lc = dynamic_cast<LCPCompressor*>(sel.compressor.get());
                if (!lc) fail("--blocks is only available for lcpcomp");
                lc->compress_blocks(inp.raw(), block_size, out, ndev);
            } else {
                if (sel.restrictions.has_restrictions()) inp = Input(inp, sel.restrictions);  // :268-270
                sel.compressor->compress(inp, out);
            }
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
            const bool container = BlockContainer::is_container(inp.raw().data(), inp.raw().size());   // blocks undo their own restrictions
            if (sel.restrictions.has_restrictions() && !container) out = Output(out, sel.restrictions);  // :336-338
            sel.compressor->decompress(inp, out);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::ofstream of(ofile, std::ios::binary | std::ios::trunc);
        if (!of) fail("Could not open " + ofile + " for writing");
        of.write((const char*)result.data(), (std::streamsize)result.size());
        if (stats) {
            // tudocomp_driver.cpp:361-391: {"meta": {...}, "data": <root phase>}; a phase is PhaseData::to_json
            // (tudocomp_stat/PhaseData.hpp:79-111): title, timeStart / timeEnd (ms), memOff / memPeak / memFinal, stats as
            // [{key, value}] (values are strings), sub phases.  Phase titles and logged keys are the reference's
            // (LCPCompressor.hpp:104-136, ArraysComp.hpp:39-68, LZSSFactors.hpp:130-131); phase times are the device times
            // of the stages laid end to end from the start of the run; the mem* fields hold device arena bytes (root only).
            const long long wall0 = (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t0.time_since_epoch()).count();
            const long long epoch = (long long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
            std::string js;
            char buf[1024];
            auto phase_open = [&](const char* title, double a, double b, unsigned long long mem) {
                std::snprintf(buf, sizeof(buf), "{\"title\":\"%s\",\"timeStart\":%lld,\"timeEnd\":%lld,\"memOff\":0,\"memPeak\":%llu,\"memFinal\":0,\"stats\":[",
                              title, wall0 + (long long)a, wall0 + (long long)b, mem);
                js += buf;
            };
            bool first_kv = true;
            auto kv = [&](const char* k, unsigned long long v) {
                std::snprintf(buf, sizeof(buf), "%s{\"key\":\"%s\",\"value\":\"%llu\"}", first_kv ? "" : ",", k, v);
                js += buf; first_kv = false;
            };
            auto subs = [&] { js += "],\"sub\":["; first_kv = true; };
            auto phase_close = [&] { js += "]}"; };
            auto* c = decompress ? nullptr : dynamic_cast<LCPCompressor*>(sel.compressor.get());
            if (c) {
                const tdc_gpu_stats& st = c->last_stats;
                double t = st.ms_h2d;
                phase_open("root", 0, ms, (unsigned long long)st.arena_bytes); subs();
                phase_open("Construct Text DS", t, t + st.ms_sa + st.ms_phi + st.ms_plcp, 0); subs();
                phase_open("Construct SA", t, t + st.ms_sa, 0); subs(); phase_close(); js += ","; t += st.ms_sa;
                phase_open("Construct Phi Array", t, t + st.ms_phi, 0); subs(); phase_close(); js += ","; t += st.ms_phi;
                phase_open("Construct PLCP Array", t, t + st.ms_plcp, 0); subs(); phase_close(); t += st.ms_plcp;
                phase_close(); js += ",";
                phase_open("Factorize", t, t + st.ms_factorize, 0);
                kv("maxlcp", st.maxlcp); kv("entries", st.entries); kv("threshold", (unsigned long long)c->threshold());
                kv("factors", st.factors);
                subs(); phase_close(); js += ","; t += st.ms_factorize;
                // "Sort Factors" has no device counterpart: factors are marks in position space
                phase_open("Flatten Factors", t, t + st.ms_flatten, 0);
                kv("num_flattened", st.num_flattened); kv("max_depth_lb", st.max_depth_lb);
                subs(); phase_close(); js += ","; t += st.ms_flatten;
                phase_open("Encode Factors", t, t + st.ms_encode, 0); subs(); phase_close();
                phase_close();
            } else {
                phase_open("root", 0, ms, 0); subs(); phase_close();
            }
            std::printf("{\"meta\":{\"title\":\"\",\"startTime\":%lld,\"config\":\"%s\",\"input\":\"%s\",\"inputSize\":%zu,\"output\":\"%s\","
                        "\"outputSize\":%zu,\"rate\":%.6f},\"data\":%s", epoch, algo.c_str(), file.c_str(), in_size, ofile.c_str(),
                        result.size(), in_size ? (double)result.size() / (double)in_size : 0.0, js.c_str());
            if (c) {                                                           // additions: wall time and the device stage times
                const tdc_gpu_stats& st = c->last_stats;
                std::printf(",\"timeTotalMs\":%.3f,\"gpu_ms\":{\"h2d\":%.3f,\"sa\":%.3f,\"phi\":%.3f,\"plcp\":%.3f,\"factorize\":%.3f,"
                            "\"flatten\":%.3f,\"encode\":%.3f,\"d2h\":%.3f,\"total\":%.3f}",
                            ms, st.ms_h2d, st.ms_sa, st.ms_phi, st.ms_plcp, st.ms_factorize, st.ms_flatten, st.ms_encode, st.ms_d2h, st.ms_total);
            }
            std::printf("}\n");
        }
        return 0;
    } catch (const std::exception& e) {
        fail(e.what());                                                                   // :392-395
    }
}
