"""MI355X-native PixPro + optical-flow hot path behind the reference's `contrast` package surface."""
