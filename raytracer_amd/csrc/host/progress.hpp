// Progress bar with ETA on stderr, as reference src/progress.rs:12-115 draws it.
#pragma once
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

namespace rt {

class TimedProgressBar {
public:
    // char_progression: UTF-8 string of glyphs from "empty" to "full" (" -=≡" or " .:", src/main.rs:201-205)
    TimedProgressBar(std::FILE* stream, size_t length, const std::string& label, const std::string& char_progression,
                     uint32_t number_samples, uint32_t times_will_be_updated);
    void update();  // src/progress.rs:48-107
    void clear();   // :109-114

private:
    std::FILE* stream_;
    size_t length_;
    std::string label_;
    std::vector<std::string> chars_;
    uint32_t number_samples_, times_will_be_updated_;
    std::vector<std::chrono::steady_clock::time_point> update_times_;
};

}  // namespace rt
