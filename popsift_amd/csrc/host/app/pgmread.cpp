// pgmread.cpp -- PGM / PPM reader of popsift-demo and popsift-match, without Boost.
//
// Behaviour of the reference's reader (src/application/pgmread.cpp:38-257), which the regression protocol relies on:
//   * P2 / P3 (ASCII) and P5 / P6 (binary); '#' comment lines between the header fields;
//   * the header is three LINES: type, "W H", maxval (one field group per line, as the reference parses it);
//   * maxval != 255 (ASCII) or > 255 (binary, two bytes per sample IN HOST BYTE ORDER, as the reference reads
//     them) is rescaled with (unsigned char)(v * 255.0 / maxval);
//   * colour is converted with OpenCV's integer weights: (4899 r + 9617 g + 1868 b) >> 14 (:25-28).
#include "pgmread.h"

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>

using namespace std;

namespace {
const uint32_t RATE_SHIFT = 14, R_RATE = 4899, G_RATE = 9617, B_RATE = 1868;

inline unsigned char to_gray( unsigned r, unsigned g, unsigned b )
{
    return (unsigned char)( ( R_RATE * r + G_RATE * g + B_RATE * b ) >> RATE_SHIFT );
}

// next header line that is not a comment, leading blanks removed; false at end of file
bool header_line( ifstream& f, string& line )
{
    for( ;; ) {
        if( !getline( f, line ) ) return false;
        size_t i = 0;
        while( i < line.size() && isspace( (unsigned char)line[i] ) ) i++;
        line.erase( 0, i );
        if( line.empty() || line[0] != '#' ) return true;
    }
}
} // namespace

unsigned char* readPGMfile( const string& filename, int& w, int& h )
{
    ifstream pgmfile( filename.c_str(), ios::binary );
    if( !pgmfile.is_open() ) {
        cerr << "File " << filename << " could not be opened for reading" << endl;
        return nullptr;
    }
    string line;
    if( !header_line( pgmfile, line ) || line.size() < 2 ) {
        cerr << "File " << filename << " is too short" << endl;
        return nullptr;
    }
    int type;
    if( line.compare( 0, 2, "P2" ) == 0 ) type = 2;
    else if( line.compare( 0, 2, "P3" ) == 0 ) type = 3;
    else if( line.compare( 0, 2, "P5" ) == 0 ) type = 5;
    else if( line.compare( 0, 2, "P6" ) == 0 ) type = 6;
    else {
        cerr << "File " << filename << " can only contain P2, P3, P5 or P6 PGM images" << endl;
        return nullptr;
    }
    if( !header_line( pgmfile, line ) ) { cerr << "File " << filename << " is too short" << endl; return nullptr; }
    if( sscanf( line.c_str(), "%d %d", &w, &h ) != 2 ) {
        cerr << "File " << filename << " PGM type header (" << type << ") must be followed by comments and WxH info" << endl
             << "but line contains " << line << endl;
        return nullptr;
    }
    if( w <= 0 || h <= 0 ) { cerr << "File " << filename << " has meaningless image size" << endl; return nullptr; }
    int maxval = 0;
    if( !header_line( pgmfile, line ) ) { cerr << "File " << filename << " is too short" << endl; return nullptr; }
    if( sscanf( line.c_str(), "%d", &maxval ) != 1 || maxval <= 0 ) {
        cerr << "File " << filename << " PGM dimensions must be followed by comments and max value info" << endl;
        return nullptr;
    }

    const size_t n = (size_t)w * h;
    unique_ptr<unsigned char[]> out( new unsigned char[n] );
    auto too_short = [&]() -> unsigned char* { cerr << "File " << filename << " file too short" << endl; return nullptr; };
    auto scale = [&]( int v ) { return maxval == 255 ? (unsigned char)v : (unsigned char)( v * 255.0 / maxval ); };

    switch( type ) {
    case 2:
        for( size_t i = 0; i < n; i++ ) {
            int v; pgmfile >> v;
            if( pgmfile.fail() ) return too_short();
            out[i] = scale( v );
        }
        break;
    case 3:
        for( size_t i = 0; i < n; i++ ) {
            int r, g, b; pgmfile >> r >> g >> b;
            if( pgmfile.fail() ) return too_short();
            out[i] = to_gray( scale( r ), scale( g ), scale( b ) );
        }
        break;
    case 5:
        if( maxval < 256 ) {
            pgmfile.read( (char*)out.get(), (streamsize)n );
            if( pgmfile.fail() ) return too_short();
        } else {
            unique_ptr<unsigned short[]> i2( new unsigned short[n] );
            pgmfile.read( (char*)i2.get(), (streamsize)( n * 2 ) );
            if( pgmfile.fail() ) return too_short();
            for( size_t i = 0; i < n; i++ ) out[i] = (unsigned char)( i2[i] * 255.0 / maxval );
        }
        break;
    default: // 6
        if( maxval < 256 ) {
            unique_ptr<unsigned char[]> i2( new unsigned char[n * 3] );
            pgmfile.read( (char*)i2.get(), (streamsize)( n * 3 ) );
            if( pgmfile.fail() ) return too_short();
            for( size_t i = 0; i < n; i++ ) out[i] = to_gray( i2[3 * i], i2[3 * i + 1], i2[3 * i + 2] );
        } else {
            unique_ptr<unsigned short[]> i2( new unsigned short[n * 3] );
            pgmfile.read( (char*)i2.get(), (streamsize)( n * 3 * 2 ) );
            if( pgmfile.fail() ) return too_short();
            for( size_t i = 0; i < n; i++ ) out[i] = to_gray( i2[3 * i], i2[3 * i + 1], i2[3 * i + 2] );
        }
        break;
    }
    return out.release();
}
