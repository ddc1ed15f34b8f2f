extern "C" void gatherer_test_stall(void);
