// pgmread.h -- PGM / PPM reader of the command line tools (reference: src/application/pgmread.h)
#pragma once
#include <string>
/// Reads a P2 / P3 / P5 / P6 file into a w*h array of bytes (new[]; the caller deletes it).  nullptr on error.
unsigned char* readPGMfile( const std::string& filename, int& w, int& h );
