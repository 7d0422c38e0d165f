"""Round 6: per-kernel durations of mi_fd_infer_images on 256 resident u8 frames (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rs_face_detection_tflite_amd as mi
fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
rs = np.random.RandomState(3)
frames = torch.from_numpy(rs.randint(0, 256, (256, 256, 256, 3)).astype(np.uint8)).cuda()
for _ in range(30):
    out = fd.infer_images(frames, cap=8)
torch.cuda.synchronize()
