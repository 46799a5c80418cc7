/* Status:: messages, as /root/reference/include_test/Status.h:28-58 */
#ifndef STATUS_H
#define STATUS_H
#include <iostream>
#include <string>
namespace Status
{
    inline void print_message(const std::string& m) { std::cout << m << std::endl; }
    inline void print_warning(const std::string& m) { std::cout << "WARNING: " << m << std::endl; }
    inline void print_error(const std::string& m) { std::cerr << "ERROR: " << m << std::endl; }
}
#endif
