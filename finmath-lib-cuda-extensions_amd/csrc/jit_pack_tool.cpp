// jit_pack_tool — build-time compiler of the kernel pack (jit.hpp): reads program and rolled-loop descriptions (csrc/kernel_pack.txt), generates
// each program's specialised-kernel source exactly as the run-time tier does, compiles it with hiprtc (no device needed) and
// stores the code object where Jit::compile looks behind the user's cache.     usage: jit_pack_tool <descriptions> <output dir>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <set>
#include <string>
#include "jit.hpp"

int main(int argc, char** argv)
{
    if (argc == 3 && std::string(argv[1]) == "--check") {            // round trip of every description: parse, describe again, compare (no compilation)
        std::ifstream in(argv[2]);
        std::string line; int n = 0, bad = 0;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            fm::DevProgramArgs proto; fm::RolledBody body;
            const std::string again = fm::jit_parse_description(line, proto) ? fm::jit_describe(proto) : fm::jit_parse_description(line, body) ? fm::jit_describe(body) : std::string("<unparsed>");
            ++n;
            if (again != line) { ++bad; std::fprintf(stderr, "round trip differs:\n  %s\n  %s\n", line.c_str(), again.c_str()); }
        }
        std::printf("%d descriptions, %d round-trip differences\n", n, bad);
        return bad ? 1 : 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--source") {           // the generated source of the kernel fm_jit_<hash> (as --names lists it), under the current environment's generator knobs
        std::ifstream in(argv[2]);
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            fm::DevProgramArgs proto; fm::RolledBody body;
            std::string source;
            if (fm::jit_parse_description(line, proto)) source = fm::jit_generate_source(proto);
            else if (fm::jit_parse_description(line, body)) source = fm::jit_generate_rolled_source(body);
            else continue;
            uint64_t h = 1469598103934665603ull;
            for (unsigned char c : source) { h ^= c; h *= 1099511628211ull; }
            char name[40]; std::snprintf(name, sizeof name, "fm_jit_%016llx", (unsigned long long)h);
            if (std::string(argv[3]) == name || std::string(argv[3]) == line.substr(0, std::strlen(argv[3]))) { std::fputs(source.c_str(), stdout); return 0; }
        }
        std::fprintf(stderr, "no such kernel\n");
        return 1;
    }
    if (argc == 3 && std::string(argv[1]) == "--names") {            // kernel name (as profilers show it) and the start of its description, one per line
        std::ifstream in(argv[2]);
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            fm::DevProgramArgs proto; fm::RolledBody body;
            std::string source;
            if (fm::jit_parse_description(line, proto)) source = fm::jit_generate_source(proto);
            else if (fm::jit_parse_description(line, body)) source = fm::jit_generate_rolled_source(body);
            else continue;
            uint64_t h = 1469598103934665603ull;
            for (unsigned char c : source) { h ^= c; h *= 1099511628211ull; }
            std::printf("fm_jit_%016llx  %s\n", (unsigned long long)h, line.c_str());
        }
        return 0;
    }
    // <descriptions> <output dir> [K N]: only the descriptions K, K + N, K + 2N, … (the build runs N tools side by side: hiprtc compiles
    // one kernel at a time per process)
    int slice = 0, slices = 1;
    if (argc == 5) { slice = std::atoi(argv[3]); slices = std::atoi(argv[4]); if (slices < 1 || slice < 0 || slice >= slices) { std::fprintf(stderr, "jit_pack_tool: bad slice\n"); return 2; } }
    else if (argc != 3) { std::fprintf(stderr, "usage: %s <descriptions> <output dir> [K N]  |  %s --check <descriptions>\n", argv[0], argv[0]); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { std::fprintf(stderr, "jit_pack_tool: cannot read %s\n", argv[1]); return 2; }
    const auto t0 = std::chrono::steady_clock::now();
    std::set<std::string> seen;
    std::string line;
    int compiled = 0, failed = 0, lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        if (line.empty() || line[0] == '#' || !seen.insert(line).second) continue;
        if ((int)((seen.size() - 1) % (size_t)slices) != slice) continue;
        fm::DevProgramArgs proto;
        fm::RolledBody body;
        std::string source, log;
        if (fm::jit_parse_description(line, proto)) source = fm::jit_generate_source(proto);
        else if (fm::jit_parse_description(line, body)) source = fm::jit_generate_rolled_source(body);
        else { std::fprintf(stderr, "jit_pack_tool: %s:%d is not a kernel description\n", argv[1], lineno); ++failed; continue; }
        if (fm::jit_precompile(source, argv[2], &log)) ++compiled;
        else { std::fprintf(stderr, "jit_pack_tool: %s:%d does not compile: %s\n", argv[1], lineno, log.c_str()); ++failed; }
    }
    std::printf("kernel pack: %d programs compiled into %s in %.1f s%s\n", compiled, argv[2], std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                failed ? " (FAILURES above)" : "");
    return failed ? 1 : 0;
}
