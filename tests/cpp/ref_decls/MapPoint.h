// Stand-in for the reference header of the same name: the mock data model, visible as ORB_SLAM3::Frame / KeyFrame / MapPoint / Map.
#pragma once
#include "mock_model_sophus.h"
namespace ORB_SLAM3 { using ::Frame; using ::KeyFrame; using ::MapPoint; using ::Map; }
