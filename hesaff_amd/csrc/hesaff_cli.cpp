// hesaff_cli.cpp -- `hesaff <image>` : same command line, stdout line and output file as
// the reference's main() (hesaff.cpp:133-180); the work runs on the MI355X through
// libhesaff_amd.so.  Input: binary PGM/PPM (P5/P6).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "hesaff.hpp"

int main(int argc, char **argv)
{
   if (argc > 1) {
      uint8_t *data = nullptr;
      int w = 0, h = 0, ch = 0;
      if (hesaff_read_pnm(argv[1], &data, &w, &h, &ch) != HESAFF_OK) {
         fprintf(stderr, "hesaff: cannot read '%s' (binary PGM/PPM with maxval 255 expected)\n", argv[1]);
         return 1;
      }
      try {
         hesaff_amd::HessianAffineParams par;
         hesaff_amd::AffineHessianDetector detector(par);
         const auto t1 = std::chrono::steady_clock::now();
         detector.detectPyramidKeypoints(data, w, h, ch);
         const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
         std::cout << "Detected " << detector.g_numberOfPoints << " keypoints and " << detector.g_numberOfAffinePoints
                   << " affine shapes in " << dt << " sec." << std::endl;
         const std::string name = std::string(argv[1]) + ".hesaff.sift";
         std::ofstream out(name.c_str());
         if (!out) { fprintf(stderr, "hesaff: cannot write '%s'\n", name.c_str()); hesaff_free(data); return 1; }
         detector.exportKeypoints(out);
      } catch (const std::exception &e) {
         fprintf(stderr, "hesaff: %s\n", e.what());
         hesaff_free(data);
         return 1;
      }
      hesaff_free(data);
   } else {
      printf("\nUsage: hesaff image_name.ppm\nDetects Hessian Affine points and describes them using SIFT descriptor.\nThe detector assumes that the vertical orientation is preserved.\n\n");
   }
   return 0;
}
