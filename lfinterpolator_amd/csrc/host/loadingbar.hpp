// loadingbar.hpp — console progress bar with the interface the reference uses from the un-vendored
// ichlubna/loadingBar submodule: LoadingBar bar(count); bar.add();  (reference src/lfLoader.cpp:60,65,
// src/interpolator.cu:103,112,131,305,314)
#pragma once

#include <cstddef>
#include <iostream>

class LoadingBar
{
    public:
        explicit LoadingBar(size_t total, std::ostream &out = std::cout) : total{total ? total : 1}, out{out} { draw(); }
        void add(size_t amount = 1)
        {
            done = done + amount > total ? total : done + amount;
            draw();
            if(done == total && !finished)
            {
                out << std::endl;
                finished = true;
            }
        }

    private:
        size_t total, done{0};
        std::ostream &out;
        bool finished{false};
        int lastCells{-1};
        void draw()
        {
            constexpr int WIDTH{40};
            const int cells = static_cast<int>(done * WIDTH / total);
            if(cells == lastCells)
                return;
            lastCells = cells;
            out << '\r' << '[';
            for(int i = 0; i < WIDTH; i++)
                out << (i < cells ? '#' : ' ');
            out << "] " << (done * 100 / total) << '%' << std::flush;
        }
};
