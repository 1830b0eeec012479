// See progress.hpp.
#include "progress.hpp"

#include <stdexcept>

namespace rt {

TimedProgressBar::TimedProgressBar(std::FILE* stream, size_t length, const std::string& label,
                                   const std::string& char_progression, uint32_t number_samples,
                                   uint32_t times_will_be_updated)
    : stream_(stream), length_(length), label_(label), number_samples_(number_samples),
      times_will_be_updated_(times_will_be_updated) {
    if (char_progression.empty()) throw std::invalid_argument("empty glyph progression");
    for (size_t i = 0; i < char_progression.size();) {  // split into UTF-8 code points (.chars())
        unsigned char c = (unsigned char)char_progression[i];
        size_t n = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
        chars_.push_back(char_progression.substr(i, n));
        i += n;
    }
    update_times_.reserve(times_will_be_updated);
}

static std::string repeat(const std::string& s, size_t n) {
    std::string out;
    for (size_t i = 0; i < n; i++) out += s;
    return out;
}

void TimedProgressBar::update() {
    update_times_.push_back(std::chrono::steady_clock::now());
    // The reference computes `self.length - self.label.len() - 15` in usize (a too-short bar panics there).
    size_t overhead = label_.size() + 15;
    size_t length = length_ > overhead ? length_ - overhead : 0;
    size_t multiplier = chars_.size() - 1;
    if (multiplier == 0) multiplier = 1;
    size_t progress = multiplier * update_times_.size() * length / (times_will_be_updated_ ? times_will_be_updated_ : 1);
    size_t total = multiplier * length;
    size_t remainder = total > progress ? total - progress : 0;
    const std::string& full_char = chars_.back();
    const std::string& empty_char = chars_.front();
    size_t progress_modulus = progress % multiplier;

    // Mean interval between the last `number_samples` updates -> time left.
    size_t n = update_times_.size();
    size_t take = n < number_samples_ ? n : number_samples_;
    uint64_t secs_left = 0;
    if (take >= 2) {
        auto span = update_times_[n - 1] - update_times_[n - take];
        double average = std::chrono::duration<double>(span).count() / (double)(take - 1);
        double left = (double)(times_will_be_updated_ > n ? times_will_be_updated_ - n : 0) * average;
        secs_left = (uint64_t)left;
    }
    char eta[32];
    std::snprintf(eta, sizeof(eta), "%2llu:%02llu:%02llu", (unsigned long long)(secs_left / 3600),
                  (unsigned long long)((secs_left % 3600) / 60), (unsigned long long)(secs_left % 60));
    std::string line = "\r" + label_ + ": [" + repeat(full_char, progress / multiplier) +
                       (progress_modulus != 0 ? chars_[progress_modulus] : std::string()) +
                       repeat(empty_char, remainder / multiplier) + "] " + eta + " ";
    std::fwrite(line.data(), 1, line.size(), stream_);
    std::fflush(stream_);
}

void TimedProgressBar::clear() {
    std::string line = "\r" + std::string(length_ + label_.size() + 12, ' ') + "\r";
    std::fwrite(line.data(), 1, line.size(), stream_);
    std::fflush(stream_);
}

}  // namespace rt
