// arguments.hpp — minimal command-line access with the call-site shape the reference's main() uses
// (reference src/main.cpp:6-12, 28, 31, 35, 39; the reference gets it from the un-vendored ichlubna/arguments submodule):
//   Arguments args(argc, argv);  std::string s = static_cast<std::string>(args["-i"]);  float f = args["-f"];
//   if(!args["-i"]) …;  args.printHelpIfPresent(text)
#pragma once

#include <cctype>
#include <iostream>
#include <string>
#include <vector>

class Arguments
{
    public:
        class Value
        {
            public:
                Value(bool present, std::string text) : present{present}, text{std::move(text)} {}
                explicit operator std::string() const { return text; }
                operator float() const
                {
                    if(text.empty())
                        return 0.0f;
                    try
                    {
                        return std::stof(text);
                    }
                    catch(const std::exception &)
                    {
                        return 0.0f;
                    }
                }
                explicit operator int() const { return static_cast<int>(static_cast<float>(*this)); }
                explicit operator bool() const { return present; }
                bool operator!() const { return !present; }

            private:
                bool present;
                std::string text;
        };

        Arguments(int argc, char **argv)
        {
            for(int i = 1; i < argc; i++)
                tokens.emplace_back(argv[i]);
        }

        // value following the flag; a flag followed by another flag (or by nothing) is present with an empty value
        Value operator[](const std::string &flag) const
        {
            for(size_t i = 0; i < tokens.size(); i++)
                if(tokens[i] == flag)
                {
                    if(i + 1 < tokens.size() && !isFlag(tokens[i + 1]))
                        return {true, tokens[i + 1]};
                    return {true, ""};
                }
            return {false, ""};
        }

        bool printHelpIfPresent(const std::string &text) const
        {
            for(const auto &t : tokens)
                if(t == "-h" || t == "--help")
                {
                    std::cout << text << std::endl;
                    return true;
                }
            return false;
        }

    private:
        std::vector<std::string> tokens;
        static bool isFlag(const std::string &t)
        {
            // "-0.5" is a value, "-f" a flag
            return t.size() >= 2 && t[0] == '-' && !(std::isdigit(static_cast<unsigned char>(t[1])) || t[1] == '.');
        }
};
