// config.h -- flat "%YAML:1.0" key/value reader standing in for the reference's cv::FileStorage
// singleton (reference include/lzb_vio/config.h, src/config.cpp).  Same public surface:
// Config::SetParameterFile(path) and Config::Get<T>(key).  A missing key returns T() like an
// empty FileNode does, but warns once.
#pragma once
#ifndef lzb_vio_CONFIG_H
#define lzb_vio_CONFIG_H

#include <map>
#include <sstream>
#include "lzb_vio/common_include.h"

namespace lzb_vio {

class Config {
private:
    static std::shared_ptr<Config> config_;
    std::map<std::string, std::string> kv_;
    Config() {}
public:
    ~Config() {}
    static bool SetParameterFile(const std::string &filename);
    static bool Has(const std::string &key);
    template <typename T>
    static T Get(const std::string &key)
    {
        T v = T();
        if (!config_) return v;
        auto it = config_->kv_.find(key);
        if (it == config_->kv_.end()) {
            LZB_LOG("WARNING", "config key '%s' missing, using default", key.c_str());
            return v;
        }
        std::istringstream ss(it->second);
        ss >> v;
        return v;
    }
};

template <>
inline std::string Config::Get<std::string>(const std::string &key)
{
    if (!config_) return std::string();
    auto it = config_->kv_.find(key);
    return it == config_->kv_.end() ? std::string() : it->second;
}

}  // namespace lzb_vio
#endif
