// main.cpp — command line of the interpolator; flags, defaults, messages and exit codes as in reference src/main.cpp:4-57
// (-i -t -o -f -r -m -s -a -h), plus -n (views), -b (benchmark runs), -d (device) and --synthetic for runs without a dataset.
#include <iostream>
#include <memory>
#include <sstream>
#include <vector>

#include "arguments.hpp"
#include "interpolator.h"

int main(int argc, char **argv)
{
    Arguments args(argc, argv);
    std::string path = static_cast<std::string>(args["-i"]);
    std::string trajectory = static_cast<std::string>(args["-t"]);
    std::string outputPath = static_cast<std::string>(args["-o"]);
    float focus = args["-f"];
    float range = args["-r"];
    std::string method = static_cast<std::string>(args["-m"]);

    std::string helpText{ "Usage:\n"
                          "Example: lfInterpolator -i /MyAmazingMachine/thoseImages -t 0.0,0.0,1.0,1.0  -o ./outputs\n"
                          "-o - output path\n"
                          "-i - folder with lf grid images - named as row_column.extension, e.g. 01_12.png\n"
                          "-t - trajectory of the camera in normalized coordinates of the grid format: startCol,startRow,endCol,endRow\n"
                          "-s - the amount of the spatial 3D effect - affects how much are views close to the virtual one prioritized (default=3.0)\n"
                          "-a - aspect ratio of the spacing of the capturing cameras in the grid (horizontal/vertical space) (default=1)\n"
                          "-m - interpolation method:\n"
                          "     STD - standard interpolation kernel (exact fp32)\n"
                          "     TEN_WM - matrix cores (fp16 MFMA)\n"
                          "The following arguments are normalized offsets of the images in shift & sum\n"
                          "-f - focusing value (default=0)\n"
                          "-r - focusing range (will be added to the focusing value) - will produce all-focused result if used\n"
                          "Additional arguments:\n"
                          "-n - number of views rendered along the trajectory (default=64)\n"
                          "-b - number of timed kernel launches (default=100)\n"
                          "-d - GPU index (default=0)\n"
                          "-g - number of GPUs: the views are split over GPUs d … d+g-1, the input grid is broadcast once (default=1)\n"
                          "-q - also store quilt.png: the first cols*rows views as cols,rows tiles (e.g. 5,9 for a Looking Glass quilt)\n"
                          "--synthetic cols,rows,width,height[,seed] - use a generated light field instead of -i\n"
                          "--unified-map - all-focus TEN_WM reads the filtered focus map like STD (the reference reads the unfiltered one)\n"
                        };
    if(args.printHelpIfPresent(helpText))
        return 0;

    float effect = static_cast<float>(args["-s"]);
    if(effect <= 0)
        effect = 3;

    float aspect = static_cast<float>(args["-a"]);
    if(aspect <= 0)
        aspect = 1;

    const bool synthetic = static_cast<bool>(args["--synthetic"]);
    if((!args["-i"] && !synthetic) || !args["-t"] || !args["-o"] || !args["-m"])
    {
        std::cerr << "Missing required parameters. Use -h for help." << std::endl;
        return EXIT_FAILURE;
    }

    try
    {
        int device = static_cast<int>(args["-d"]);
        std::unique_ptr<Interpolator> interpolator;
        if(synthetic)
        {
            std::stringstream spec(static_cast<std::string>(args["--synthetic"]));
            std::string token;
            std::vector<long> numbers;
            while(std::getline(spec, token, ','))
                numbers.push_back(std::stol(token));
            if(numbers.size() < 4)
                throw std::runtime_error("--synthetic expects cols,rows,width,height[,seed]");
            interpolator = std::make_unique<Interpolator>(lfi::IVec2{int(numbers[0]), int(numbers[1])}, lfi::IVec2{int(numbers[2]), int(numbers[3])},
                                                          numbers.size() > 4 ? uint32_t(numbers[4]) : 0x1F1Fu, device);
        }
        else
        {
            Interpolator::setDefaultDevice(device);
            interpolator = std::make_unique<Interpolator>(path);
        }
        if(args["-n"])
            interpolator->setViewCount(static_cast<int>(args["-n"]));
        if(args["-g"])
            interpolator->setGpuCount(static_cast<int>(args["-g"]));
        if(args["-b"])
            interpolator->setBenchmarkRuns(static_cast<size_t>(static_cast<int>(args["-b"])));
        if(args["--unified-map"])
            interpolator->setUnifiedFocusMap(true);
        if(args["-q"])
        {
            std::stringstream spec(static_cast<std::string>(args["-q"]));
            std::string a, b;
            if(!std::getline(spec, a, ',') || !std::getline(spec, b, ','))
                throw std::runtime_error("-q expects cols,rows");
            interpolator->setQuilt({std::stoi(a), std::stoi(b)});
        }
        interpolator->interpolate(outputPath, trajectory, focus, range, method, effect, aspect);
    }
    catch(const std::exception &e)
    {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
