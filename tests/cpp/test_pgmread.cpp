// test_pgmread.cpp -- drives the tools' PGM/PPM reader: prints "w h" and writes the decoded bytes to argv[2]
#include "pgmread.h"
#include <cstdio>
#include <fstream>
int main( int argc, char** argv )
{
    if( argc < 3 ) return 2;
    int w = 0, h = 0;
    unsigned char* d = readPGMfile( argv[1], w, h );
    if( d == nullptr ) { printf( "FAILED\n" ); return 1; }
    printf( "%d %d\n", w, h );
    std::ofstream of( argv[2], std::ios::binary );
    of.write( (const char*)d, (std::streamsize)w * h );
    delete[] d;
    return 0;
}
