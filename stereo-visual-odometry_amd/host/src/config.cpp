// config.cpp -- flat YAML reader ("%YAML:1.0", '#' comments, "key: value", dotted keys, any bytes
// in comments) replacing cv::FileStorage (reference src/config.cpp:5-18).
#include "lzb_vio/config.h"
#include <fstream>

namespace lzb_vio {

static std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

bool Config::SetParameterFile(const std::string &filename)
{
    if (config_ == nullptr) config_ = std::shared_ptr<Config>(new Config);
    config_->kv_.clear();
    std::ifstream f(filename.c_str());
    if (!f.is_open()) {
        LZB_LOG("ERROR", "parameter file %s does not exist.", filename.c_str());
        return false;
    }
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '%') continue;             // %YAML:1.0
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
        if (key.empty()) continue;
        if (val.size() >= 2 && (val[0] == '"' || val[0] == '\'') && val.back() == val[0]) val = val.substr(1, val.size() - 2);
        config_->kv_[key] = val;
    }
    return true;
}

bool Config::Has(const std::string &key) { return config_ && config_->kv_.count(key) > 0; }

std::shared_ptr<Config> Config::config_ = nullptr;

}  // namespace lzb_vio
