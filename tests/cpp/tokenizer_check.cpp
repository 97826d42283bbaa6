// Test driver: prints whisper_mi::Tokenizer(argv[1]).decode(ids from argv[2..]) — compared with the Python mirror.
#include <cstdlib>
#include <iostream>

#include "whisper_mi.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    try {
        whisper_mi::Tokenizer tok(argv[1]);
        std::vector<int> ids;
        for (int i = 2; i < argc; ++i) ids.push_back(std::atoi(argv[i]));
        std::cout << tok.size() << "\n" << tok.decode(ids);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
