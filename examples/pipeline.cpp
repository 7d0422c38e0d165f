// C++ counterpart of the reference's end-to-end test (/root/reference/src/lib.rs:18-40, README.md:27-46):
//   detector -> faces[0] -> face_detection_to_roi -> FaceLandmark -> iris_roi_from_face_landmarks -> IrisLandmark x2.
// usage: pipeline <raw-rgb-file> <width> <height> [model_dir]       (raw file = height*width*3 bytes, RGB)
//        pipeline <picture.jpg> 0 0 [model_dir]                     (encoded bytes -> convert_image_to_mat, utils.rs:8-21, as the
//                                                                    reference's test does with include_bytes!(man.jpg))
// Prints one line per result so that tests can parse it.
#include <cmath>
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
    int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const std::string dir = argc > 4 ? argv[4] : "./models";
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<std::uint8_t> px((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const bool encoded = w == 0 && h == 0;
    if (!encoded && px.size() != static_cast<std::size_t>(w) * h * 3) {
        std::fprintf(stderr, "raw image has %zu bytes, expected %d\n", px.size(), w * h * 3);
        return 2;
    }
    try {
        using namespace mi_face;
        OwnedImage decoded;
        if (encoded) {
            decoded = convert_image_to_mat(px.data(), px.size());
            w = decoded.width; h = decoded.height;
            std::printf("decoded %d %d\n", w, h);
        }
        const Image image = encoded ? decoded.image() : Image{px.data(), w, h, 3 * w};
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
        // iris_landmark.rs:380-398: mesh point 33 (left eye corner) becomes the first left contour landmark, 263 the right one
        const auto refined = update_face_landmarks_with_iris_results(lmks, left, right);
        std::printf("refined %zu lm33 %.9g %.9g lm263 %.9g %.9g contour0 %.9g %.9g %.9g %.9g\n", refined.size(), refined[33].x, refined[33].y,
                    refined[263].x, refined[263].y, left.contour[0].x, left.contour[0].y, right.contour[0].x, right.contour[0].y);
        // the batched u8 entry (mi_fd_infer_images): the same picture three times in one call gives the single-image result three times
        {
            std::vector<std::uint8_t> three;
            for (int k = 0; k < 3; k++)
                for (int y = 0; y < h; y++) three.insert(three.end(), image.rgb + static_cast<std::size_t>(y) * image.stride, image.rgb + static_cast<std::size_t>(y) * image.stride + 3 * w);
            const auto batch = face_detection.infer_batch(three.data(), 3, w, h, 3 * w);
            bool same = batch.size() == 3;
            for (const auto& f : batch) same = same && f.size() == faces.size() && std::fabs(f[0].bbox().xmin - bb.xmin) < 2e-5 && std::fabs(f[0].score - faces[0].score) < 2e-5;
            std::printf("batch3 %s\n", same ? "same" : "DIFFERENT");
        }
        const auto eye_box = bbox_from_landmarks(left.contour);
        std::printf("left_eye_box %.9g %.9g %.9g %.9g\n", eye_box[0], eye_box[1], eye_box[2], eye_box[3]);
    } catch (const mi_face::Error& e) {
        std::fprintf(stderr, "mi_face error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
