// C++ counterpart of the reference's end-to-end test (/root/reference/src/lib.rs:18-40, README.md:27-46):
//   detector -> faces[0] -> face_detection_to_roi -> FaceLandmark -> iris_roi_from_face_landmarks -> IrisLandmark x2.
// usage: pipeline <raw-rgb-file> <width> <height> [model_dir]       (raw file = height*width*3 bytes, RGB)
// Prints one line per result so that tests can parse it.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "mi_face.hpp"

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <raw-rgb> <width> <height> [model_dir]\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const std::string dir = argc > 4 ? argv[4] : "./models";
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<std::uint8_t> px((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (px.size() != static_cast<std::size_t>(w) * h * 3) {
        std::fprintf(stderr, "raw image has %zu bytes, expected %d\n", px.size(), w * h * 3);
        return 2;
    }
    try {
        using namespace mi_face;
        const Image image{px.data(), w, h, 3 * w};
        FaceDetection face_detection(FaceDetectionModel::BackCamera, dir);
        const auto faces = face_detection.infer(image, std::nullopt);
        std::printf("faces %zu\n", faces.size());
        if (faces.empty()) return 0;
        const BBox bb = faces[0].bbox();
        std::printf("bbox %.9g %.9g %.9g %.9g score %.9g\n", bb.xmin, bb.ymin, bb.xmax, bb.ymax, faces[0].score);
        const Rect face_roi = face_detection_to_roi(faces[0], {w, h});
        std::printf("face_roi %.17g %.17g %.17g %.17g %.17g\n", face_roi.x_center, face_roi.y_center, face_roi.width, face_roi.height, face_roi.rotation);
        FaceLandmark face_landmark(dir + "/face_landmark.tflite");
        const auto lmks = face_landmark.infer(image, face_roi);
        std::printf("landmarks %zu\n", lmks.size());
        if (lmks.empty()) return 0;
        std::printf("lm0 %.9g %.9g %.9g lm467 %.9g %.9g %.9g\n", lmks[0].x, lmks[0].y, lmks[0].z, lmks[467].x, lmks[467].y, lmks[467].z);
        const auto [left_eye_roi, right_eye_roi] = iris_roi_from_face_landmarks(lmks, {w, h});
        IrisLandmark iris_landmark(dir + "/iris_landmark.tflite");
        const IrisResults right = iris_landmark.infer(image, right_eye_roi, true);
        const IrisResults left = iris_landmark.infer(image, left_eye_roi, false);
        std::printf("right_iris_center %.9g %.9g left_iris_center %.9g %.9g contour %zu iris %zu\n", right.iris[0].x, right.iris[0].y,
                    left.iris[0].x, left.iris[0].y, left.eyeball_contour().size(), left.iris.size());
    } catch (const mi_face::Error& e) {
        std::fprintf(stderr, "mi_face error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
