// prove — stdin REPL worker with the protocol of the reference CLI (src/main.rs:121-186):
//   > prove --witness W --zkey Z --proof P --public Q --device HIP
// prints COMMAND_COMPLETED after every command, COMMAND_EMPTY for blank lines, COMMAND_EXIT on "exit".
//   > verify --proof P --public Q --vk verification_key.json
#include <iostream>
#include <sstream>
#include <string>

#include "groth16_prover.h"

static void print_help()
{
  std::cout << "Usage:\n  prove [--system groth16] --witness <file> --zkey <file> --proof <file> --public <file> --device <HIP>\n  verify [--system groth16] --proof <file> --public <file> --vk <file>\n  exit\n";
}

int main()
{
  Groth16CacheManager* cm = groth16_cache_manager_new();
  {
    // the device the first `prove` will most likely name ("HIP" → device 0, or the first of ICICLE_SNARK_DEVICES): its streams
    // and staging buffers are created while the worker waits for its first command
    int ids[1] = {0};
    if (groth16_parse_device("HIP", ids, 1) >= 1) groth16_cache_manager_prewarm(cm, ids[0]);
  }
  std::string line;
  for (;;) {
    std::cout << "> " << std::flush;
    if (!std::getline(std::cin, line)) break;
    std::istringstream in(line);
    std::string cmd;
    if (!(in >> cmd)) {
      std::cout << "COMMAND_EMPTY\nCOMMAND_COMPLETED" << std::endl;
      continue;
    }
    if (cmd == "exit" || cmd == "EXIT" || cmd == "Exit") {
      std::cout << "COMMAND_EXIT\nCOMMAND_COMPLETED" << std::endl;
      break;
    }
    if (cmd == "prove") {
      // defaults of src/main.rs:46-50, except the device: this build registers "HIP" ("CUDA" is an alias)
      std::string witness = "witness.wtns", zkey = "circuit_final.zkey", proof = "proof.json", pub = "public.json", device = "CUDA", a, v;
      bool ok = true;
      while (in >> a) {
        if (a == "--system") {
          if (in >> v && v != "groth16" && v != "Groth16" && v != "GROTH16") {
            std::cerr << "Unknown proof system: " << v << std::endl;
            ok = false;
          }
        } else if (a == "--witness") in >> witness;
        else if (a == "--zkey") in >> zkey;
        else if (a == "--proof") in >> proof;
        else if (a == "--public") in >> pub;
        else if (a == "--device") in >> device;
        else print_help();
      }
      if (!ok) {
        print_help();
        continue;
      }
      int rc = groth16_prove(witness.c_str(), zkey.c_str(), proof.c_str(), pub.c_str(), device.c_str(), cm);
      if (rc != 0) {
        // the reference unwraps (aborts) here; report and keep the worker alive instead
        std::cerr << "prove failed (" << rc << "): " << groth16_last_error() << std::endl;
      }
      std::cout << "COMMAND_COMPLETED" << std::endl;
    } else if (cmd == "verify") {
      // defaults of src/main.rs:84-86
      std::string proof = "proof.json", pub = "public.json", vk = "verification_key.json", a, v;
      bool ok = true;
      while (in >> a) {
        if (a == "--system") {
          if (in >> v && v != "groth16" && v != "Groth16" && v != "GROTH16") {
            std::cerr << "Unknown proof system: " << v << std::endl;
            ok = false;
          }
        } else if (a == "--proof") in >> proof;
        else if (a == "--public") in >> pub;
        else if (a == "--vk") in >> vk;
        else print_help();
      }
      if (!ok) {
        print_help();
        continue;
      }
      int rc = groth16_verify(proof.c_str(), pub.c_str(), vk.c_str());
      // the reference panics on a rejected proof (assert, src/lib.rs:79); report and keep the worker alive instead
      if (rc == 0) std::cout << "VERIFY_OK" << std::endl;
      else {
        std::cerr << "verify failed (" << rc << "): " << groth16_verify_last_error() << std::endl;
        std::cout << "VERIFY_FAILED" << std::endl;
      }
      std::cout << "COMMAND_COMPLETED" << std::endl;
    } else {
      print_help();
    }
  }
  std::cout << "Exiting CLI worker..." << std::endl;
  groth16_cache_manager_free(cm);
  return 0;
}
