// cli_args.h — minimal long/short option parser for the two drop-in tools (the reference uses
// cxxopts, an empty submodule in its tree). Accepts "--name value", "--name=value", "-x value"
// and boolean flags, like the invocations kmers_gwas.py builds (kmers_gwas.py:133-148).
#pragma once
#include "env.h"
using kgwas::opt_str;
using kgwas::opt_int;
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct CliOption {
    std::string long_name;
    char short_name;   // 0 = none
    bool takes_value;
    std::string help, def;  // def shown in help only
};

class CliArgs {
   public:
    explicit CliArgs(std::vector<CliOption> opts) : opts_(std::move(opts)) {}

    void parse(int argc, char** argv) {
        for (int i = 1; i < argc; i++) {
            std::string a(argv[i]);
            const CliOption* o = nullptr;
            std::string inline_val;
            bool has_inline = false;
            if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
                std::string name = a.substr(2);
                size_t eq = name.find('=');
                if (eq != std::string::npos) {
                    inline_val = name.substr(eq + 1);
                    name = name.substr(0, eq);
                    has_inline = true;
                }
                for (auto& x : opts_)
                    if (x.long_name == name) o = &x;
                if (!o) throw std::invalid_argument("Option '" + name + "' does not exist");
            } else if (a.size() == 2 && a[0] == '-') {
                for (auto& x : opts_)
                    if (x.short_name && x.short_name == a[1]) o = &x;
                if (!o) throw std::invalid_argument("Option '" + a.substr(1) + "' does not exist");
            } else {
                throw std::invalid_argument("Unexpected argument '" + a + "'");
            }
            if (o->takes_value) {
                if (has_inline)
                    vals_[o->long_name] = inline_val;
                else {
                    if (i + 1 >= argc) throw std::invalid_argument("Option '" + o->long_name + "' is missing an argument");
                    vals_[o->long_name] = argv[++i];
                }
            } else {
                vals_[o->long_name] = "true";
            }
        }
    }
    size_t count(const std::string& n) const { return vals_.count(n); }
    std::string str(const std::string& n) const {
        auto it = vals_.find(n);
        if (it == vals_.end()) throw std::invalid_argument("Option '" + n + "' not present");
        return it->second;
    }
    std::string str(const std::string& n, const std::string& def) const { return count(n) ? str(n) : def; }
    unsigned long long u64(const std::string& n) const { return parse_u64(n, str(n)); }
    unsigned long long u64(const std::string& n, unsigned long long def) const { return count(n) ? u64(n) : def; }
    double f64(const std::string& n) const {
        try {
            size_t pos = 0;
            std::string s = str(n);
            double v = std::stod(s, &pos);
            if (pos != s.size()) throw std::invalid_argument("");
            return v;
        } catch (const std::invalid_argument&) {
            throw std::invalid_argument("Argument '" + str(n) + "' failed to parse");
        }
    }
    double f64(const std::string& n, double def) const { return count(n) ? f64(n) : def; }

    std::string help(const std::string& prog, const std::string& desc) const {
        std::string h = desc + "\nUsage:\n  " + prog + " [OPTION...]\n\n";
        for (auto& o : opts_) {
            std::string l = "  ";
            l += o.short_name ? (std::string("-") + o.short_name + ", ") : std::string("    ");
            l += "--" + o.long_name + (o.takes_value ? " arg" : "");
            while (l.size() < 34) l += ' ';
            l += o.help + (o.def.empty() ? "" : " (default: " + o.def + ")");
            h += l + "\n";
        }
        return h;
    }

   private:
    static unsigned long long parse_u64(const std::string& n, const std::string& s) {
        try {
            size_t pos = 0;
            if (!s.empty() && s[0] == '-') throw std::invalid_argument("");
            unsigned long long v = std::stoull(s, &pos);
            if (pos != s.size()) throw std::invalid_argument("");
            return v;
        } catch (const std::exception&) {
            throw std::invalid_argument("Argument '" + s + "' failed to parse");
        }
    }
    std::vector<CliOption> opts_;
    std::map<std::string, std::string> vals_;
};

// Every output of the tool is written and closed: leave without unpinning gigabytes of host memory, freeing device
// buffers and tearing the HIP runtime down one object at a time - the kernel reclaims all of it at process exit, and
// for `associate_kmers` on a 40 M-row table the orderly way was 0.35 s of a 0.88 s run (0.19 s in the session's
// destructors, the rest in the runtime's exit handlers). KGWAS_CLI_FULL_TEARDOWN=1 returns instead (leak checkers).
// A result that could not be written (a full disk, a closed pipe behind `emma_kinship_kmers`' matrix) is an error, not exit 0.
inline void cli_finish() {
    std::cout.flush();
    std::cerr.flush();
    const bool ok = std::cout.good() && fflush(nullptr) == 0;
    if (!ok) {
        fputs("error: writing the output failed\n", stderr);
        _exit(1);
    }
    if (!opt_str("KGWAS_CLI_FULL_TEARDOWN")) _exit(0);
}
